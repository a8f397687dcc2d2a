"""Development aid: where does KLNMF.fit() spend host time at config c2?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import salamander_amd as sal
from salamander_amd import synthetic as orc
V, N, K = 96, 100000, 50
X, W0, H0 = orc.synthetic_problem(V, N, K, seed=0)
def t(label, f):
    t0 = time.perf_counter(); r = f(); print(f"{label:40s} {1e3*(time.perf_counter()-t0):8.2f} ms"); return r
e = t("Engine() first (module load)", lambda: sal.Engine(N, V, K))
t("upload_X (76.8 MB)", lambda: e.upload_X(X))
t("upload_H (40 MB)", lambda: e.upload_H(H0))
t("upload_W", lambda: e.upload_W(W0))
t("first kl_step(1)+sync", lambda: (e.kl_step(1), e.sync()))
t("objective", lambda: e.objective())
t("download_H", lambda: e.download_H())
e.close()
e = t("Engine() second", lambda: sal.Engine(N, V, K)); e.close()
for rep in range(2):
    adata = sal.AnnData(X.copy())
    m = sal.models.KLNMF(K, "custom", min_iterations=500, max_iterations=500)
    t0 = time.perf_counter()
    m._setup_adata(adata); t1 = time.perf_counter()
    m._initialize(None, {"signatures_mat": W0.copy(), "exposures_mat": H0.copy()}); t2 = time.perf_counter()
    m._setup_fitting_parameters(None); m._sync_to_device(); t3 = time.perf_counter()
    m._device_objective(); m._device_steps(500, None); m._engine.sync(); t4 = time.perf_counter()
    m._sync_from_device(); t5 = time.perf_counter()
    print(f"fit pieces #{rep}: setup_adata {1e3*(t1-t0):.1f} ms, initialize {1e3*(t2-t1):.1f} ms, sync_to_device {1e3*(t3-t2):.1f} ms, 500 steps {1e3*(t4-t3):.1f} ms, sync_from_device {1e3*(t5-t4):.1f} ms")
