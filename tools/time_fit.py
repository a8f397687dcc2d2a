"""Development aid: wall-clock of repeated KLNMF(50).fit(adata) calls with the default init at c2, with the time of the
pieces (engine creation + ingest + init, the 500-step loop, result download)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import salamander_amd as sal
from salamander_amd.synthetic import synthetic_problem

V, N, K = 96, 100000, 50
X, _, _ = synthetic_problem(V, N, K, seed=0)
for rep in range(8):
    adata = sal.AnnData(X.copy())
    model = sal.models.KLNMF(K, min_iterations=500, max_iterations=500)
    t0 = time.perf_counter()
    model._setup_adata(adata)
    t1 = time.perf_counter()
    model._initialize(None, None)
    t2 = time.perf_counter()
    adata2 = sal.AnnData(X.copy())
    model2 = sal.models.KLNMF(K, min_iterations=500, max_iterations=500)
    t3 = time.perf_counter()
    if rep == 3:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable(); model2.fit(adata2); pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(8)
    else:
        model2.fit(adata2)
    t4 = time.perf_counter()
    print(f"rep {rep}: setup {1e3*(t1-t0):.1f} ms, init {1e3*(t2-t1):.1f} ms | fit end to end {1e3*(t4-t3):.1f} ms", flush=True)

import cProfile, pstats
adata = sal.AnnData(X.copy())
model = sal.models.KLNMF(K, min_iterations=500, max_iterations=500)
pr = cProfile.Profile(); pr.enable(); model.fit(adata); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
