"""KL step time at c2, c3's shard, K = 30 and small sizes: median of 9 blocks of 200 steps (one line per size)."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from salamander_amd import Engine, _lib, synthetic
out = []
for N, K in [(100000, 50), (125000, 50), (100000, 30), (100000, 40), (20000, 50), (1000000, 50)]:
    X, W0, H0 = synthetic.synthetic_problem(96, N, K, seed=0)
    e = Engine(N, 96, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    steps = 200 if N <= 200000 else 40
    e.kl_step(300 if N <= 200000 else 40); e.sync()
    blocks = []
    for _ in range(9):
        t0 = time.perf_counter(); e.kl_step(steps); e.sync()
        blocks.append((time.perf_counter() - t0) / steps * 1e6)
    out.append(f"N={N} K={K}: {statistics.median(blocks):.2f}")
    e.close()
print(os.path.basename(_lib.LIB_PATH), " | ".join(out), flush=True)
