"""Development aid: us per joint KLNMF step at 96 x 100 000 for signature counts on both sides of the 64-signature chunk."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import salamander_amd as sal
from salamander_amd.synthetic import synthetic_problem

N, V = 100000, 96
for K in (50, 64, 65, 100, 128, 200):
    X, W0, H0 = synthetic_problem(V, N, K, seed=0)
    e = sal.Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.kl_step(5); e.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); e.kl_step(50); e.sync(); ts.append((time.perf_counter() - t0) / 50 * 1e6)
    t0 = time.perf_counter(); e.objective(); to = (time.perf_counter() - t0) * 1e6
    flops = 6.0 * V * K * N
    print(f"K={K:4d}: {np.median(ts):8.1f} us/step  ({flops / np.median(ts) / 1e6 / 78.6:.3f} of the fp64 MFMA peak on 6VKN), objective {to:7.1f} us")
    e.close()
