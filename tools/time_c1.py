"""Development aid: config c1 (the reference's own CPU-runnable case): KLNMF(5) / MvNMF(5) on the PCAWG breast catalogue
(96 x ~200), default settings -- wall clock of fit() on the device, where every launch is shorter than its dispatch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
import salamander_amd as sal

df = pd.read_csv(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pcawg_breast_sbs.csv"), index_col=0)
X = df.T.values.astype(float) if df.shape[0] == 96 else df.values.astype(float)
print("samples x features:", X.shape)
for cls in (sal.models.KLNMF, sal.models.MvNMF):
    for rep in range(3):
        m = cls(5, max_iterations=2000)
        t0 = time.perf_counter(); m.fit(sal.AnnData(X.copy())); dt = time.perf_counter() - t0
    print(f"{cls.__name__}: {m.n_iterations_} iterations in {dt * 1e3:.1f} ms = {dt / m.n_iterations_ * 1e6:.1f} us per iteration; objective {m.history['objective_function'][-1]:.6g}")
