"""Development aid: KL step time and fused-kernel roofline fraction at several shard sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from salamander_amd import synthetic as orc
from salamander_amd import Engine
V, K = 96, 50
for N in (100000, 125000, 250000, 1000000):
    X, W0, H0 = orc.synthetic_problem(V, min(N, 250000), K, seed=0)
    if N > X.shape[0]:
        reps = N // X.shape[0]; X = np.tile(X, (reps, 1)); H0 = np.tile(H0, (reps, 1))
    e = Engine(N, V, K); e.upload_X(X); e.upload_W(W0); e.upload_H(H0)
    e.kl_step(10); e.sync()
    tot, fused, tail = e.profile_kl_steps(100, 0, 8)
    print(f"N={N}: {tot/100*1e3:.1f} us/step, fused {fused*1e3:.1f} us = {6*V*K*N/(fused*1e-3)/78.6e12*100:.1f}% of fp64 MFMA peak, tail {tail*1e3:.1f} us; forward+objective {e.profile_objective(5)*1e3:.1f} us")
    e.close()
