"""MvNMF steps on sample-sharded engines, queued form against the classic one: `world` ranks as processes on ONE GPU
(peer-to-peer exchange, control plane gloo), N samples in total.  us per step (max over ranks), median of 7 blocks of 50.

    python tools/time_sharded_mv.py [world] [N] [K]
"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch.multiprocessing as mp


def worker(rank, world, port, N, K):
    import torch.distributed as dist
    from salamander_amd import Engine, synthetic
    from salamander_amd.distributed import attach_peer_exchange, shard_bounds
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, W0, H0 = synthetic.synthetic_problem(96, N, K, seed=2)
    a, b = shard_bounds(N, world, rank)
    e = Engine(b - a, 96, K)
    e.upload_X(X[a:b])
    attach_peer_exchange(e)
    res = {}
    for queued in (True, False):
        e.set_mv_queued(queued)
        e.upload_W(W0), e.upload_H(H0[a:b])
        g = e.mv_step(20, 0, 1.0, 1.0, 1.0); e.sync()
        blocks = []
        for _ in range(7):
            dist.barrier()
            t0 = time.perf_counter(); g, f = e.mv_step_objective(50, 0, 1.0, 1.0, g, more_follows=True); e.sync()
            blocks.append((time.perf_counter() - t0) / 50 * 1e6)
        res[queued] = statistics.median(blocks)
    out = [None] * world
    dist.all_gather_object(out, res)
    if rank == 0:
        q, c = max(r[True] for r in out), max(r[False] for r in out)
        print(f"world={world} N={N} ({b - a} per rank) K={K}: queued {q:7.1f} us/step, classic {c:7.1f} us/step ({c / q:.2f}x)", flush=True)
    dist.barrier()
    e.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 30
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(world, port, N, K), nprocs=world, join=True)
