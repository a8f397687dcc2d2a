"""c4 (MvNMF, 96 x 100 000, K = 30) queued steps and the K = 30 joint KL step with the objective folded in: us per step,
median of 9 blocks.  Prints one line; used by A/B runs of two builds (SALNMF_LIB=... python tools/time_mv_c4.py)."""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from salamander_amd import Engine, _lib, synthetic
N, K = 100000, 30
X, W0, H0 = synthetic.synthetic_problem(96, N, K, seed=2)
e = Engine(N, 96, K)
e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
g = e.mv_step(300, 0, 1.0, 1.0, 1.0); e.sync()
blocks = []
for _ in range(9):
    t0 = time.perf_counter(); g = e.mv_step(50, 0, 1.0, 1.0, g); e.sync()
    blocks.append((time.perf_counter() - t0) / 50 * 1e6)
mv = statistics.median(blocks)
f = e.mv_objective(1.0, 1.0)
e.upload_W(W0), e.upload_H(H0)
e.kl_step(100); e.sync()
blocks = []
for _ in range(9):
    t0 = time.perf_counter()
    for i in range(20):
        e.kl_step_objective(1 + i, 10, 0)
    e.sync()
    blocks.append((time.perf_counter() - t0) / 200 * 1e6)
print(f"{os.path.basename(_lib.LIB_PATH):18s} c4 MvNMF queued {mv:7.2f} us/step (objective after {f!r}); K=30 KL blocks of 10 steps with the objective folded in {statistics.median(blocks):7.2f} us/step", flush=True)
e.close()
