"""Development aid: us per joint KLNMF step at cohort sizes of real catalogues (200 ... 30 000 samples), per-step launches against
the persistent multi-step kernel (only in builds made with SALNMF_WITH_PERSISTENT=1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, salamander_amd as sal
from salamander_amd import _lib
from salamander_amd.synthetic import synthetic_problem
has_p = bool(_lib.load().salnmf_build_flags() & _lib.BUILD_PERSISTENT)
print("persistent kernel in this build:", has_p)
for N, K in ((200, 5), (2780, 20), (10000, 30), (30000, 50)):
    X, W0, H0 = synthetic_problem(96, N, K, seed=1)
    row = []
    for persistent in ((False, True) if has_p else (False,)):
        e = sal.Engine(N, 96, K)
        e.set_persistent(persistent) if persistent else None
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
        e.kl_step(64); e.sync()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); e.kl_step(640); e.sync(); ts.append((time.perf_counter() - t0) / 640 * 1e6)
        row.append(np.median(ts))
        e.close()
    print(f"N={N} K={K}: " + ", ".join(f"{'persistent' if i else 'per-step launches'} {t:.1f} us/step" for i, t in enumerate(row)))
