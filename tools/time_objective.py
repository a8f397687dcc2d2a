"""Average duration of the objective's forward pass (forward_kernel<KS, 0>, HIP events around the launches) and of the
W@H-only pass at c2 (96 x 100 000, K = 50) and c4 (K = 30)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from salamander_amd import Engine, synthetic
for K in (50, 30):
    X, W0, H0 = synthetic.synthetic_problem(96, 100000, K, seed=0)
    e = Engine(100000, 96, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.kl_step(5, 0)
    e.profile_objective(20)
    obj = [e.profile_objective(200) * 1e3 for _ in range(3)]
    rec = [e.profile_reconstruct(200) * 1e3 for _ in range(3)]
    print(f"K={K}: objective pass {['%.2f' % v for v in obj]} us, W@H pass {['%.2f' % v for v in rec]} us", flush=True)
    e.close()
