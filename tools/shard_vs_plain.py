"""Development aid: the 125 000-sample shard of c3 as a plain engine and as a one-rank sharded engine (peer exchange attached):
step time in long blocks and the per-dispatch times, to see what the sharded path itself costs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import salamander_amd as sal
from salamander_amd.synthetic import synthetic_problem

N, V, K = int(sys.argv[1]) if len(sys.argv) > 1 else 125000, 96, 50
X, W0, H0 = synthetic_problem(V, N, K, seed=0)


def block_us(e, n=200, reps=7):
    e.kl_step(50); e.sync()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); e.kl_step(n); e.sync(); ts.append((time.perf_counter() - t0) / n * 1e6)
    return float(np.median(ts))


for sharded in (False, True, False, True):
    e = sal.Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    if sharded:
        h = e.p2p_export(1)
        e.p2p_connect(0, [h], N)
    us = block_us(e)
    if sharded:
        tl = e.profile_sharded_steps(200, 0)
        detail = " ".join(f"{k} {v:.2f}" for k, v in tl.items())
    else:
        total, fused, tail = e.profile_kl_steps(400, 0, 4)
        detail = f"fused {fused * 1e3:.2f} tail {tail * 1e3:.2f}"
    print(f"sharded={sharded}: {us:.2f} us/step | {detail}", flush=True)
    e.close()
