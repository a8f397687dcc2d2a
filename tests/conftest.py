import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_FIX = os.path.join(GOLDEN, "ref_fixtures")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # libsalnmf.so is git-ignored: cross-compile it when it is missing OR older than any of its sources
    # (build() compares mtimes and is a no-op otherwise), so the suites never run against a stale binary
    if os.path.exists("/opt/rocm/bin/hipcc"):
        import __graft_entry__

        __graft_entry__.build()


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped only where there is provably no HIP device; a missing or
    unloadable extension on a GPU box is a hard failure, never a silent skip."""
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if not gpu_items:
        return
    from salamander_amd import _lib

    if os.path.exists(_lib.LIB_PATH) and _lib.load().salnmf_device_count() == 0 and not os.path.exists("/dev/kfd"):
        skip = pytest.mark.skip(reason="no HIP device in this container")
        for it in gpu_items:
            it.add_marker(skip)


def rel_l2(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.fixture(scope="session")
def golden():
    class G:
        small = np.load(os.path.join(GOLDEN, "kl_small.npz"))
        synth = np.load(os.path.join(GOLDEN, "kl_synth.npz"))
        pcawg = np.load(os.path.join(GOLDEN, "kl_pcawg.npz"))
        mv = np.load(os.path.join(GOLDEN, "mv_synth.npz"))

    return G


def read_counts(path):
    import pandas as pd

    return pd.read_csv(path, index_col=0)
