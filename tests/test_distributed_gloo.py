"""The N>1 path on CPU: world_size-2 ``gloo`` run of the sharded step driver.

Each rank owns a contiguous block of samples (``shard_bounds``), runs the split step
(partial -> all-reduce of the (K, V) numerator -> identical W tail) through
``salamander_amd.distributed.host_collective_steps`` with an oracle-backed shard engine
(tests only), and the result is compared with the unsharded oracle.
"""

import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import GOLDEN, ROOT, rel_l2
from salamander_amd.distributed import shard_bounds


def test_shard_bounds_cover_and_align():
    for n, w in ((100000, 8), (1000000, 8), (10, 2), (17, 4), (16, 8), (1, 3), (125, 2)):
        blocks = [shard_bounds(n, w, r) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        for (a0, a1), (b0, b1) in zip(blocks[:-1], blocks[1:]):
            assert a1 == b0 and a0 <= a1
            assert a1 % 16 == 0 or a1 == n  # no 16-sample tile straddles two ranks
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 16
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, case, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from _fake_engine import FakeShardEngine
    from salamander_amd.distributed import broadcast_from_rank0, host_collective_steps

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = np.load(os.path.join(GOLDEN, "kl_synth.npz"))
        X, W0, H0 = g["X"].T, g["W0"].T, g["H0"].T  # sample-major (N, V), (K, V), (N, K)
        wkl = g["wkl"] if case["wkl"] else None
        wlh = g["wlh"] if case["wlh"] else None
        a, b = shard_bounds(X.shape[0], world, rank)
        e = FakeShardEngine(b - a, X.shape[1], W0.shape[0])
        e.upload_X(X[a:b])
        # rank 1 starts from a perturbed W to prove the broadcast makes the ranks identical
        e.upload_W(broadcast_from_rank0(W0 if rank == 0 else W0 * 1.5))
        e.upload_H(H0[a:b])
        e.set_weights(None if wkl is None else wkl[a:b], None if wlh is None else wlh[a:b])
        host_collective_steps(e, case["steps"], case["given"])
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), W=e.W, H=e.H, a=a, b=b)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize(
    "case",
    [
        dict(steps=3, given=0, wkl=False, wlh=False),
        dict(steps=2, given=7, wkl=True, wlh=True),
        dict(steps=2, given=50, wkl=False, wlh=False),
    ],
)
def test_two_rank_sharded_steps_match_unsharded_oracle(tmp_path, case):
    from oracle import klnmf_oracle as orc

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    g = np.load(os.path.join(GOLDEN, "kl_synth.npz"))
    X, W, H = g["X"], g["W0"], g["H0"]
    wkl = g["wkl"] if case["wkl"] else None
    wlh = g["wlh"] if case["wlh"] else None
    for _ in range(case["steps"]):
        W, H = orc.update_WH(X, W, H, wkl, wlh, case["given"])
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    assert np.array_equal(parts[0]["W"], parts[1]["W"])  # bit-identical W on every rank
    assert rel_l2(parts[0]["W"], W.T) < 1e-12
    Hs = np.concatenate([p["H"] for p in parts], axis=0)
    assert rel_l2(Hs, H.T) < 1e-12
