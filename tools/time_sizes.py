"""us per KLNMF / MvNMF step (device-resident, blocks of 200 / 50 steps) at cohort sizes users of the reference run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from salamander_amd import Engine, synthetic
SIZES = [(192, 5), (500, 5), (500, 10), (1000, 10), (2780, 10), (2780, 20), (5000, 10), (10000, 20), (20000, 20), (50000, 30)]
if len(sys.argv) > 1:
    SIZES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for N, K in SIZES:
    X, W0, H0 = synthetic.synthetic_problem(96, N, K, seed=0)
    e = Engine(N, 96, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.kl_step(50, 0); e.sync()
    kl = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); e.kl_step(200, 0); e.sync(); kl = min(kl, (time.perf_counter() - t0) / 200 * 1e6)
    e.upload_W(W0), e.upload_H(H0)
    g = e.mv_step(10, 0, 1.0, 1.0, 1.0); e.sync()
    mv = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); g, f = e.mv_step_objective(50, 0, 1.0, 1.0, g, more_follows=True); e.sync(); mv = min(mv, (time.perf_counter() - t0) / 50 * 1e6)
    e.close()
    print(f"N={N:6d} K={K:2d}: KLNMF {kl:7.2f} us/step, MvNMF {mv:7.2f} us/step", flush=True)
