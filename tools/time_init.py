import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import salamander_amd as sal
from salamander_amd.synthetic import synthetic_problem
from salamander_amd.device_init import initialize_on_device
V, N, K = 96, 100000, 50
X, _, _ = synthetic_problem(V, N, K, seed=0)
def T(label, f):
    t0 = time.perf_counter(); r = f(); dt = time.perf_counter() - t0
    print(f"  {label}: {dt*1e3:.1f} ms"); return r
for rep in range(4):
    print("rep", rep)
    e = T("Engine()", lambda: sal.Engine(N, V, K))
    T("upload_X", lambda: e.upload_X(X))
    T("sync", lambda: e.sync())
    G = T("init_gram", lambda: e.init_gram())
    T("eigh", lambda: np.linalg.eigh(G[0]))
    T("initialize_on_device (all)", lambda: initialize_on_device(e, K, "nndsvd", None, N))
    T("download_H", lambda: e.download_H())
    T("close", lambda: e.close())
