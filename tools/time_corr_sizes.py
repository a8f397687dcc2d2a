"""ms per CorrNMFDet update (device-resident) at cohort sizes below c5: where the single-kernel / lockstep / batched forms are used."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import salamander_amd as sal
from salamander_amd.models import CorrNMFDet
from salamander_amd.synthetic import synthetic_problem

SIZES = [(500, 5, 5), (2000, 10, 10), (5000, 10, 10), (20000, 10, 10), (20000, 20, 20), (50000, 10, 10), (50000, 30, 30)]
if len(sys.argv) > 1:
    SIZES = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for N, K, dim in SIZES:
    X, _, _ = synthetic_problem(96, N, K, seed=1)
    np.random.seed(0)
    m = CorrNMFDet(n_signatures=K, dim_embeddings=dim, init_method="random")
    m._setup_adata(sal.AnnData(X))
    m._initialize(None, {"seed": 0})
    m._sync_to_device()
    e = m._engine
    steps = []
    for i in range(6):
        e.sync(); t0 = time.perf_counter(); m._device_steps(1, None); e.sync()
        steps.append((time.perf_counter() - t0) * 1e3)
    def timed(fn):
        e.sync(); t0 = time.perf_counter(); fn(); e.sync()
        return (time.perf_counter() - t0) * 1e3
    t_sig = timed(lambda: e.corr_update_signature_embeddings(m.variance, 0))
    t_smp = timed(lambda: e.corr_update_sample_embeddings(m.variance, 3))
    print(f"N={N:6d} K={K:2d} dim={dim:2d}: update {sorted(steps)[len(steps) // 2]:7.2f} ms (signature solves {t_sig:6.2f}, sample solves {t_smp:6.2f})", flush=True)
