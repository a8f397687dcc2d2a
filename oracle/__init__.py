"""CPU oracle for the KL-NMF hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``salamander_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and there only as the checker / reported baseline.
"""
