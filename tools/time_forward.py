"""Forward kernels at c2 and K = 30: W@H + objective terms, W@H alone (us, mean of 50 launches by HIP events) and
the blocking objective() call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from salamander_amd import Engine, _lib, synthetic
out = []
for N, K in [(100000, 50), (100000, 30), (125000, 50), (20000, 50)]:
    X, W0, H0 = synthetic.synthetic_problem(96, N, K, seed=0)
    e = Engine(N, 96, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.kl_step(100); e.sync()
    a, b = e.profile_objective(50) * 1e3, e.profile_reconstruct(50) * 1e3
    out.append(f"N={N} K={K}: objective {a:.2f} WH {b:.2f} (obj {e.objective():.6e})")
    e.close()
print(os.path.basename(_lib.LIB_PATH), " | ".join(out), flush=True)
