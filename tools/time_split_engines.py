"""Development aid: us per joint KLNMF step (and per MvNMF step) of engines with feature blocks (> 96 features), signature chunks
(> 64 signatures) and both, at 100 000 samples -- next to the one-block, one-chunk engine of c2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import salamander_amd as sal
from salamander_amd.synthetic import synthetic_problem

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
for V, K in ((96, 50), (288, 50), (96, 100), (288, 100), (1536, 30), (1536, 80)):
    X, W0, H0 = synthetic_problem(V, N, K, seed=0)
    e = sal.Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.kl_step(5); e.sync()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); e.kl_step(20); e.sync(); ts.append((time.perf_counter() - t0) / 20 * 1e6)
    t0 = time.perf_counter(); e.objective(); to = (time.perf_counter() - t0) * 1e6
    g, tm = 1.0, []
    for _ in range(3):
        t0 = time.perf_counter(); g = e.mv_step(5, 0, 1.0, 1.0, g); e.sync(); tm.append((time.perf_counter() - t0) / 5 * 1e6)
    flops = 6.0 * V * K * N
    print(f"V={V:5d} K={K:4d}: KLNMF {np.median(ts):9.1f} us/step ({flops / np.median(ts) / 1e6 / 78.6:.3f} of the fp64 MFMA peak on 6VKN), "
          f"objective {to:8.1f} us, MvNMF {np.median(tm):9.1f} us/step", flush=True)
    e.close()
