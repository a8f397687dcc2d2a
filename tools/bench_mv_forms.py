"""MvNMF step time, queued form (tail + one MVJ pass per step, line-search decision on the device) against the classic
form (two passes per step, the host decides every step), at c4 and other shapes; blocks of 50 steps, median of 9."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from salamander_amd import Engine, synthetic
for N, K in [(100000, 30), (100000, 50), (17003, 33), (5000, 10), (1000000, 30)]:
    X, W0, H0 = synthetic.synthetic_problem(96, N, K, seed=2)
    row = []
    for queued in (False, True):
        e = Engine(N, 96, K)
        e.set_mv_queued(queued)
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
        g = e.mv_step(10, 0, 1.0, 1.0, 1.0); e.sync()
        blocks = []
        for _ in range(9):
            t0 = time.perf_counter(); g, f = e.mv_step_objective(50, 0, 1.0, 1.0, g, more_follows=True); e.sync()
            blocks.append((time.perf_counter() - t0) / 50 * 1e6)
        row.append(sorted(blocks)[4])
        e.close()
    flops = 12.0 * 96 * K * N
    print(f"N={N:8d} K={K:2d}: classic {row[0]:7.1f} us/step, queued {row[1]:7.1f} us/step ({row[0] / row[1]:.2f}x; "
          f"{flops / row[1] * 1e-6 / 78.6:.3f} of the fp64 MFMA peak on 12 V K N)", flush=True)
