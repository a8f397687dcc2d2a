"""Development aid: the signature selection of init_method="separableNMF" (K deflation rounds) at c2 -- device
(Engine.init_separable on the resident X) against the host loop the reference runs (methods.py:112-135)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from salamander_amd import Engine
from salamander_amd.synthetic import synthetic_problem

V, K = 96, 50
for N in (100000, 1000000):
    X, _, _ = synthetic_problem(V, N, K, seed=0) if N == 100000 else (np.random.default_rng(0).poisson(20.0, size=(N, V)).astype(float).clip(1.1920928955078125e-07), None, None)
    e = Engine(N, V, K)
    e.upload_X(X)
    e.init_separable(K)  # warm (scratch allocation)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); chosen = e.init_separable(K); ts.append(time.perf_counter() - t0)
    e.close()
    line = f"N={N} K={K}: device selection {sorted(ts)[1]*1e3:.2f} ms"
    if N == 100000:
        t0 = time.perf_counter()
        R = X.T / X.T.sum(axis=0)
        host = []
        for _ in range(K):
            norms = (R**2).sum(axis=0); j = int(np.argmax(norms)); u = R[:, j]; R = R - np.outer(u, u @ R) / norms[j]; host.append(j)
        th = time.perf_counter() - t0
        line += f" | host loop {th:.2f} s | same indices: {host == chosen.tolist()}"
    print(line, flush=True)
