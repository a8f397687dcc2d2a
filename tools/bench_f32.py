import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import salamander_amd as sal
from salamander_amd.synthetic import synthetic_problem
V, N, K = 96, 100000, 50
X, W0, H0 = synthetic_problem(V, N, K, seed=0)
e = sal.Engine(N, V, K); e.upload_X(X); e.upload_W(W0); e.upload_H(H0); e.set_precision("f32")
e.kl_step(100); e.sync()
b = []
for _ in range(9):
    t0 = time.perf_counter(); e.kl_step(200); e.sync(); b.append((time.perf_counter() - t0) / 200)
print(f"f32 fast mode c2: {statistics.median(b)*1e6:.2f} us/step (min {min(b)*1e6:.2f})")
