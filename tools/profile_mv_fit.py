"""Development aid: where the time of MvNMF(30).fit at c4 (96 x 100 000, 500 iterations) goes (cProfile of the host side)."""
import sys, time, cProfile, pstats, io, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import salamander_amd as sal
from salamander_amd.synthetic import synthetic_problem
X, W0, H0 = synthetic_problem(96, 100000, 30, seed=2)
for rep in range(3):
    adata = sal.AnnData(X.copy())
    m = sal.models.MvNMF(30, "custom", lam=1.0, delta=1.0, min_iterations=500, max_iterations=500)
    kw = {"signatures_mat": W0.copy(), "exposures_mat": H0.copy()}
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable(); m.fit(adata, init_kwargs=kw); pr.disable()
    print("fit seconds", time.perf_counter() - t0)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:5000])
e = m._engine
g = 1.0
for n in (1, 2, 5, 10, 50):
    e.mv_step(n, 0, 1.0, 1.0, g); e.sync()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); g, f = e.mv_step_objective(n, 0, 1.0, 1.0, g); ts.append(time.perf_counter() - t0)
    print(f"mv_step_objective({n}): {np.median(ts) * 1e6:8.1f} us per call = {np.median(ts) * 1e6 / n:6.1f} us per step")
