"""us per KL step of small cohorts: the one-workgroup multi-step kernel (csrc/salnmf_small.hip) against the per-step path
(two launches per step), wall clock around 2 000 queued steps.  c1 = 192 samples, 5 signatures."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from salamander_amd import Engine, synthetic
for N, K in [(192, 5), (64, 5), (128, 5), (256, 5), (256, 16), (320, 5), (512, 5), (768, 5), (1024, 5), (1024, 16)]:
    X, W0, H0 = synthetic.synthetic_problem(96, N, K, seed=0)
    row = []
    for tiles in (0, 64):
        e = Engine(N, 96, K)
        e.set_small_cohort_tiles(tiles)
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
        e.kl_step(50, 0); e.sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); e.kl_step(2000, 0); e.sync()
            best = min(best, (time.perf_counter() - t0) / 2000 * 1e6)
        row.append(best)
        e.close()
    print(f"N={N:5d} K={K:2d}: per-step path {row[0]:6.2f} us/step, one workgroup {row[1]:6.2f} us/step  ({row[0] / row[1]:.2f}x)", flush=True)
