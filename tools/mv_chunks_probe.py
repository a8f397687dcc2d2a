"""Development aid: MvNMF steps on signature chunks (96 x 100 000, K = 100) for a kernel trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import salamander_amd as sal
from salamander_amd.synthetic import synthetic_problem
N, V, K = 100000, 96, int(sys.argv[1]) if len(sys.argv) > 1 else 100
X, W0, H0 = synthetic_problem(V, N, K, seed=0)
e = sal.Engine(N, V, K)
e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
g = e.mv_step(3, 0, 1.0, 1.0, 1.0); e.sync()
t0 = time.perf_counter(); g = e.mv_step(20, 0, 1.0, 1.0, g); e.sync()
print(f"K={K}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per MvNMF step", flush=True)
e.close()
