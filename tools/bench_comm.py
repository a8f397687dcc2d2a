"""Development aid: per-step cost of the in-engine RCCL path at world size 1 (one GPU)."""
import os, sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torch.distributed as dist
from salamander_amd import synthetic as orc
from salamander_amd import Engine
from salamander_amd.distributed import attach_communicator
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
dist.init_process_group("gloo", rank=0, world_size=1)
V, N, K = 96, 100000, 50
X, W0, H0 = orc.synthetic_problem(V, N, K, seed=0)
for with_comm in (False, True):
    e = Engine(N, V, K)
    if with_comm: attach_communicator(e)
    e.upload_X(X); e.upload_W(W0); e.upload_H(H0)
    e.kl_step(20); e.sync()
    t0 = time.perf_counter(); e.kl_step(500); e.sync(); dt = time.perf_counter() - t0
    tot, fused, tail = e.profile_kl_steps(200, 0, 8)
    print(f"comm={with_comm}: {dt/500*1e6:.1f} us/step plain; profiled: fused {fused*1e3:.1f} us, tail(+reduce+allreduce) {tail*1e3:.1f} us")
    e.close()
dist.destroy_process_group()
