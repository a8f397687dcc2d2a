"""Development aid: time MvNMF steps (config c4) and the objective kernels on the GPU."""
import sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from salamander_amd import synthetic as orc
from salamander_amd import Engine
V, N, K = 96, 100000, 30
X, W0, H0 = orc.synthetic_problem(V, N, K, seed=2)
e = Engine(N, V, K); e.upload_X(X); e.upload_W(W0); e.upload_H(H0)
g = e.mv_step(10, 0, 1.0, 1.0, 1.0); e.sync()
blocks = []
for _ in range(9):  # the protocol of bench.py: extra.c4_mvnmf (median of blocks of 50 steps), a few more blocks
    t0 = time.perf_counter(); g = e.mv_step(50, 0, 1.0, 1.0, g); e.sync(); blocks.append((time.perf_counter() - t0) / 50)
dt = sorted(blocks)[len(blocks) // 2]
print(f"c4 MvNMF K=30 96x100k: {dt*1e6:.1f} us/step median of {len(blocks)} blocks of 50 (min {min(blocks)*1e6:.1f}, max {max(blocks)*1e6:.1f}; {1/dt:.0f} steps/s), gamma={g}")
t0 = time.perf_counter()
for _ in range(20): o = e.mv_objective(1.0, 1.0)
dt = time.perf_counter() - t0
print(f"mv_objective: {dt/20*1e6:.1f} us per call (host round trip included)")
print(f"forward+objective kernel K=30: {e.profile_objective(20)*1e3:.1f} us")
e2 = Engine(N, V, 50); X2, W2, H2 = orc.synthetic_problem(V, N, 50, seed=0); e2.upload_X(X2); e2.upload_W(W2); e2.upload_H(H2)
e2.kl_step(5); e2.sync()
t0 = time.perf_counter(); e2.kl_step(500); e2.sync(); dt = time.perf_counter() - t0
print(f"c2 plain kl_step (no events): {dt/500*1e6:.1f} us/step")
