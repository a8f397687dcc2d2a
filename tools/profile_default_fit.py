import sys, time, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
import numpy as np
import salamander_amd as sal
from salamander_amd.synthetic import synthetic_problem
X, _, _ = synthetic_problem(96, 100000, 50, seed=0)
warm = sal.models.KLNMF(50, min_iterations=1, max_iterations=1)
warm.fit(sal.AnnData(X[:2000].copy()))
for rep in range(3):
    adata = sal.AnnData(X.copy())
    m = sal.models.KLNMF(50, min_iterations=500, max_iterations=500)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable(); m.fit(adata); pr.disable()
    print("fit seconds", time.perf_counter() - t0)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:9000])
