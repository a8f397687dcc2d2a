"""Dry run of tests/test_gpu_real_anndata.py on a box without anndata / mudata: the two modules are stood in for by the
package's own containers, so that the tests' own logic (shapes, names, comparisons) is exercised end to end."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from salamander_amd.anndata_compat import MiniAnnData, MiniMuData
ad = types.ModuleType("anndata"); ad.AnnData = MiniAnnData
md = types.ModuleType("mudata"); md.MuData = MiniMuData
sys.modules.setdefault("anndata", ad); sys.modules.setdefault("mudata", md)
import test_gpu_real_anndata as t
for name in [n for n in dir(t) if n.startswith("test_")]:
    getattr(t, name)()
    print("ok", name, flush=True)
