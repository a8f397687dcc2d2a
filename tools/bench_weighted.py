"""Development aid: step time of the weighted instantiation of the joint KL step at c2 (weights_kl; weights_kl + weights_lhalf;
weights_lhalf only) next to the unweighted step."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from salamander_amd import Engine
from salamander_amd.synthetic import synthetic_problem

V, N, K = 96, 100000, 50
X, W0, H0 = synthetic_problem(V, N, K, seed=0)
rng = np.random.default_rng(1)
wk, wl = rng.uniform(0.5, 2.0, N), rng.uniform(0.0, 0.2, N)
for name, w in (("unweighted", (None, None)), ("weights_kl", (wk, None)), ("weights_kl + weights_lhalf", (wk, wl)), ("weights_lhalf", (None, wl))):
    e = Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.set_weights(*w)
    e.kl_step(50); e.sync()
    blocks = []
    for _ in range(7):
        t0 = time.perf_counter(); e.kl_step(200); e.sync(); blocks.append((time.perf_counter() - t0) / 200)
    print(f"{name:28s}: {statistics.median(blocks) * 1e6:.1f} us/step (min {min(blocks) * 1e6:.1f})", flush=True)
    e.close()
