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


# ---------------------------------------------------------------------------------------------------------------
# Sample-sharded CorrNMF (config c5's split): the models with ``distributed=True`` on two gloo ranks, each with an
# oracle-backed engine that performs the engine's exchanges (tests/_fake_engine.py::FakeCorrShardEngine), against the
# unsharded oracle.  What is tested: which quantities cross ranks (numerator, scaling sums, Poisson term, sum of
# squares, ONE gather for the signature-embedding solves), the global sample count in the variance / ELBO, and that
# the replicated parameters stay bit-identical on every rank.


def _corr_problem(seed=3, N=26, V=12, K=3, dim=2):
    rng = np.random.default_rng(seed)
    X = rng.poisson(rng.gamma(2.0, 6.0, size=(N, V))).astype(float).clip(1e-7)
    W = rng.dirichlet(np.ones(V), size=K)
    beta = rng.normal(0, 0.1, K)
    alpha = np.log(X.sum(axis=1) / K)
    L = rng.normal(0, 0.3, (K, dim))
    U = rng.normal(0, 0.3, (N, dim))
    return X, W, beta, alpha, L, U


def _corr_worker(rank, world, port, out_dir, n_updates):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import salamander_amd as sal
    from _fake_engine import FakeCorrShardEngine
    from salamander_amd.models import signature_nmf

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        signature_nmf.Engine = FakeCorrShardEngine
        X, W, beta, alpha, L, U = _corr_problem()
        a, b = shard_bounds(X.shape[0], world, rank)
        adata = sal.AnnData(X[a:b].copy())
        adata.obs["scalings"] = alpha[a:b].copy()
        adata.obsm["embeddings"] = U[a:b].copy()
        # rank 1 starts from perturbed replicated parameters: the broadcasts must make the ranks identical
        bump = 1.0 if rank == 0 else 1.3
        sigs = sal.AnnData(W.copy() * bump)
        sigs.obs["scalings"] = beta.copy() * bump
        sigs.obsm["embeddings"] = L.copy() * bump
        m = sal.models.CorrNMFDet(n_signatures=W.shape[0], dim_embeddings=L.shape[1], distributed=True)
        m.adata, m.asignatures, m.variance = adata, sigs, 1.0 * bump
        m._sync_to_device()  # (the exposures are recomputed on the device at the start of every update)
        m._device_steps(n_updates, None)
        elbo = m._device_objective()
        m._sync_from_device()
        np.savez(
            os.path.join(out_dir, f"corr_rank{rank}.npz"), W=sigs.X, beta=sigs.obs["scalings"].values, L=sigs.obsm["embeddings"],
            alpha=adata.obs["scalings"].values, U=adata.obsm["embeddings"], H=adata.obsm["exposures"], variance=m.variance, elbo=elbo,
        )
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_corrnmf_det_matches_unsharded_oracle(tmp_path):
    from oracle import corrnmf_oracle as corr

    world, n_updates = 2, 2
    mp.spawn(_corr_worker, args=(world, _free_port(), str(tmp_path), n_updates), nprocs=world, join=True)
    X, W, beta, alpha, L, U = _corr_problem()
    variance = 1.0
    for _ in range(n_updates):
        W, beta, alpha, L, U, variance, H = corr.corrnmf_det_step(X, W, beta, alpha, L, U, variance)
    elbo = corr.elbo_corrnmf(X, W, H, L, U, variance)
    parts = [np.load(os.path.join(tmp_path, f"corr_rank{r}.npz")) for r in range(world)]
    for key in ("W", "beta", "L", "variance", "elbo"):  # replicated: the same bits on every rank
        assert np.array_equal(parts[0][key], parts[1][key]), key
    p0 = parts[0]
    assert rel_l2(p0["W"], W) < 1e-9 and rel_l2(p0["beta"], beta) < 1e-9
    assert np.allclose(p0["L"], L, rtol=1e-6, atol=1e-9)  # Newton-CG: same problem, summation order differs
    assert np.isclose(float(p0["variance"]), variance, rtol=1e-8)
    assert np.isclose(float(p0["elbo"]), elbo, rtol=1e-9)
    for key, want in (("alpha", alpha), ("U", U), ("H", H)):  # sharded: concatenated in rank order
        got = np.concatenate([p[key] for p in parts], axis=0)
        assert np.allclose(got, want, rtol=1e-6, atol=1e-9), key


def _mm_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import salamander_amd as sal
    from _fake_engine import FakeCorrShardEngine
    from salamander_amd.models import mmcorrnmf

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mmcorrnmf.Engine = FakeCorrShardEngine
        X1, W1, b1, a1, L1, U = _corr_problem(seed=5, V=12, K=3)
        X2, W2, b2, a2, L2, _ = _corr_problem(seed=6, V=9, K=2)
        lo, hi = shard_bounds(X1.shape[0], world, rank)
        ads = {}
        for name, X, alpha in (("mod1", X1, a1), ("mod2", X2, a2)):
            ad = sal.AnnData(X[lo:hi].copy())
            ad.obs["scalings"] = alpha[lo:hi].copy()
            ads[name] = ad
        mdata = sal.MuData(ads)
        mdata.obsm["embeddings"] = U[lo:hi].copy()
        m = sal.models.MultimodalCorrNMF([3, 2], dim_embeddings=2, distributed=True)
        m.mdata = mdata
        for name, W, beta, L in (("mod1", W1, b1, L1), ("mod2", W2, b2, L2)):
            s = sal.AnnData(W.copy())
            s.obs["scalings"] = beta.copy()
            s.obsm["embeddings"] = L.copy()
            m.asignatures[name] = s
        m.variance = 1.0
        m._sync_to_device()
        m._device_steps(2, None)
        elbo = m._device_objective()
        m._sync_from_device()
        np.savez(
            os.path.join(out_dir, f"mm_rank{rank}.npz"), W1=m.asignatures["mod1"].X, W2=m.asignatures["mod2"].X,
            L1=m.asignatures["mod1"].obsm["embeddings"], L2=m.asignatures["mod2"].obsm["embeddings"],
            U=m.mdata.obsm["embeddings"], variance=m.variance, elbo=elbo,
        )
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_multimodal_corrnmf_matches_unsharded_oracle(tmp_path):
    from oracle import corrnmf_oracle as corr

    world = 2
    mp.spawn(_mm_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    X1, W1, b1, a1, L1, U = _corr_problem(seed=5, V=12, K=3)
    X2, W2, b2, a2, L2, _ = _corr_problem(seed=6, V=9, K=2)
    Xs, Ws, betas, alphas, Ls, variance = [X1, X2], [W1, W2], [b1, b2], [a1, a2], [L1, L2], 1.0
    for _ in range(2):
        Ws, betas, alphas, Ls, U, variance, Hs = corr.mm_step(Xs, Ws, betas, alphas, Ls, U, variance)
    elbo = corr.mm_elbo(Xs, Ws, Hs, Ls, U, variance)
    parts = [np.load(os.path.join(tmp_path, f"mm_rank{r}.npz")) for r in range(world)]
    for key in ("W1", "W2", "L1", "L2", "variance", "elbo"):
        assert np.array_equal(parts[0][key], parts[1][key]), key
    p0 = parts[0]
    assert rel_l2(p0["W1"], Ws[0]) < 1e-9 and rel_l2(p0["W2"], Ws[1]) < 1e-9
    assert np.allclose(p0["L1"], Ls[0], rtol=1e-6, atol=1e-9) and np.allclose(p0["L2"], Ls[1], rtol=1e-6, atol=1e-9)
    assert np.isclose(float(p0["variance"]), variance, rtol=1e-8) and np.isclose(float(p0["elbo"]), elbo, rtol=1e-9)
    assert np.allclose(np.concatenate([p["U"] for p in parts], axis=0), U, rtol=1e-6, atol=1e-9)
