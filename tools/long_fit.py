"""Development aid: KLNMF / MvNMF fits of 3000 iterations at c2 / c4 size -- monotone objectives, finite factors, wall clock."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, salamander_amd as sal
from salamander_amd.synthetic import synthetic_problem
X, _, _ = synthetic_problem(96, 100000, 50, seed=0)
for cls, K in ((sal.models.KLNMF, 50), (sal.models.MvNMF, 30)):
    m = cls(K, max_iterations=3000)
    t = time.perf_counter(); m.fit(sal.AnnData(X.copy())); dt = time.perf_counter() - t
    h = m.history["objective_function"]
    print(cls.__name__, "iterations", m.n_iterations_, "seconds", round(dt, 3), "objective first/last", h[0], h[-1], "monotone", bool(np.all(np.diff(h) <= 1e-9 * np.abs(h[:-1]))), "finite", bool(np.isfinite(m.asignatures.X).all() and np.isfinite(m.adata.obsm["exposures"]).all()))
