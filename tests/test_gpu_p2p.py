"""The peer-to-peer exchange of the sharded steps (``salnmf_p2p_kernels.h``) with several ranks.

A one-GPU box cannot hold two RCCL ranks, but hipIpc handles work between processes on the same device: the ranks here
are separate processes with one engine each on ``cuda:0``, connected by ``attach_peer_exchange`` (handles through a
``gloo`` group), with NO RCCL communicator.  That exercises the whole protocol -- export / map, parity slots, flags,
rank-ordered sums, the waits -- except the xGMI hop itself.  Compared with the unsharded oracle on the same inputs.
"""

import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, rel_l2
from test_distributed_gloo import _free_port

pytestmark = pytest.mark.gpu

V, N, K = 96, 2500, 50
LAM, DELTA = 2.0, 0.5


def _worker(rank, world, port, out_dir, n_kl, n_mv):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from oracle import klnmf_oracle as orc
    from salamander_amd.distributed import attach_peer_exchange, shard_bounds
    from salamander_amd.engine import Engine

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, W0, H0 = orc.synthetic_problem(V, N, K, seed=5)  # sample-major: (N, V), (K, V), (N, K)
        a, b = shard_bounds(N, world, rank)
        e = Engine(b - a, V, K)
        e.upload_X(X[a:b]), e.upload_W(W0), e.upload_H(H0[a:b])
        attach_peer_exchange(e)
        assert e.comm_info() == (world, rank, N)
        # the default initialisation on the shards: Gram matrix and column norms go through the same exchange
        from salamander_amd.device_init import initialize_on_device

        S_init = initialize_on_device(e, K, "nndsvd", None, N)
        H_init = e.download_H()
        e.upload_W(W0), e.upload_H(H0[a:b])
        e.kl_step(n_kl - 4, 0)
        # four of the steps through the profiling entry point: the same launches with events and in-kernel stamps -- the
        # state must not notice, and every phase of the exchange must have taken a sane time on every rank
        tl = e.profile_sharded_steps(4, 0)
        assert all(np.isfinite(v) and v >= 0.0 for v in tl.values()) and tl["step"] > 0 and tl["longest_flag_wait"] < 5e6, tl
        assert tl["fused_pass"] + tl["tail_exchange_launch"] <= 1.5 * tl["step"] + 50, tl
        Wk, Hk = e.download_W(), e.download_H()
        obj = e.objective()  # scalar all-reduce through the same exchange
        gamma = e.mv_step(n_mv, 0, LAM, DELTA, 1.0)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), Wk=Wk, Hk=Hk, obj=obj, Wm=e.download_W(), Hm=e.download_H(), gamma=gamma, a=a, b=b,
                 S_init=S_init, H_init=H_init)
        dist.barrier()  # nobody frees its inbox while a peer may still be inside an exchange
        e.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_steps_over_peer_exchange_match_unsharded_oracle(tmp_path, world):
    from oracle import klnmf_oracle as orc

    n_kl, n_mv = 12, 4
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), n_kl, n_mv), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=5)
    W, H = W0.T, H0.T
    for _ in range(n_kl):
        W, H = orc.update_WH(X.T, W, H, None, None, 0)
    for p in parts[1:]:
        assert np.array_equal(parts[0]["Wk"], p["Wk"])  # rank-ordered sums: the same bits on every rank
        assert p["obj"] == parts[0]["obj"]
    assert rel_l2(parts[0]["Wk"], W.T) < 1e-11
    assert rel_l2(np.concatenate([p["Hk"] for p in parts], axis=0), H.T) < 1e-11
    assert np.isclose(float(parts[0]["obj"]), orc.kl_divergence(X.T, W, H), rtol=1e-10)
    # sharded nndsvd init == the same init on one engine holding all samples (sums in a different order)
    from salamander_amd.device_init import initialize_on_device
    from salamander_amd.engine import Engine

    e1 = Engine(N, V, K)
    e1.upload_X(X)
    S1 = initialize_on_device(e1, K, "nndsvd", None, N)
    for p in parts[1:]:
        assert np.array_equal(parts[0]["S_init"], p["S_init"])
    assert rel_l2(parts[0]["S_init"], S1) < 1e-9
    assert rel_l2(np.concatenate([p["H_init"] for p in parts], axis=0), e1.download_H()) < 1e-9
    e1.close()
    gamma = 1.0
    for _ in range(n_mv):
        W, H, gamma = orc.mvnmf_step(X.T, W, H, LAM, DELTA, gamma, 0)
    for p in parts[1:]:
        assert np.array_equal(parts[0]["Wm"], p["Wm"])
    assert np.isclose(float(parts[0]["gamma"]), gamma, rtol=1e-12)
    assert rel_l2(parts[0]["Wm"], W.T) < 1e-8
    assert rel_l2(np.concatenate([p["Hm"] for p in parts], axis=0), H.T) < 1e-8


def _stalled_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from oracle import klnmf_oracle as orc
    from salamander_amd.distributed import attach_peer_exchange, shard_bounds
    from salamander_amd.engine import Engine

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, W0, H0 = orc.synthetic_problem(V, 640, 8, seed=6)
        a, b = shard_bounds(640, world, rank)
        e = Engine(b - a, V, 8)
        e.upload_X(X[a:b]), e.upload_W(W0), e.upload_H(H0[a:b])
        attach_peer_exchange(e, timeout_ms=300)
        e.kl_step(2, 0)
        e.download_W()
        message = ""
        if rank == 0:  # rank 1 never joins the third exchange
            e.kl_step(3, 0)  # three exchanges: one waits 0.3 s, the next two return at once
            try:
                e.download_W()
            except RuntimeError as exc:
                message = str(exc)
        open(os.path.join(out_dir, f"msg{rank}.txt"), "w").write(message)
        dist.barrier()
        e.close()
    finally:
        dist.destroy_process_group()


def test_exchange_gives_up_when_a_rank_stays_away(tmp_path):
    import time

    t0 = time.time()
    mp.spawn(_stalled_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert "gave up waiting for another rank" in open(os.path.join(tmp_path, "msg0.txt")).read()
    assert time.time() - t0 < 60


def test_peer_exchange_argument_checks():
    from salamander_amd.engine import Engine

    e = Engine(64, 96, 5)
    with pytest.raises(RuntimeError, match="export first"):
        e.p2p_connect(0, [b"\0" * 64], 64)
    with pytest.raises(RuntimeError, match="1..8 ranks"):
        e.p2p_export(9)
    with pytest.raises(RuntimeError, match="not connected"):
        e.set_p2p(True)
    h = e.p2p_export(1)
    assert len(h) == 64
    with pytest.raises(RuntimeError, match="exported already"):
        e.p2p_export(1)
    with pytest.raises(RuntimeError, match="does not match"):
        e.p2p_connect(0, [h, h], 64)
    # a single rank: the exchange degenerates to a copy through its own inbox
    X = np.random.default_rng(0).poisson(3.0, (64, 96)).astype(float)
    W0 = np.random.default_rng(1).random((5, 96)) + 0.1
    H0 = np.random.default_rng(2).random((64, 5)) + 0.1
    ref = Engine(64, 96, 5)
    for eng in (e, ref):
        eng.upload_X(X), eng.upload_W(W0), eng.upload_H(H0)
    e.p2p_connect(0, [h], 64)
    e.kl_step(3, 0), ref.kl_step(3, 0)
    assert np.array_equal(e.download_W(), ref.download_W()) and np.array_equal(e.download_H(), ref.download_H())
    # the per-phase timeline of the sharded step (world of one rank: no peer to wait for); same state afterwards
    tl = e.profile_sharded_steps(5, 0)
    ref.kl_step(5, 0)
    assert np.array_equal(e.download_W(), ref.download_W()) and np.array_equal(e.download_H(), ref.download_H())
    assert set(tl) == {"step", "fused_pass", "tail_exchange_launch", "local_slab_reduce", "peer_stores_and_flags", "wait_for_peer_flags",
                       "read_and_sum_peer_rows", "w_row_finish", "longest_flag_wait"}
    # (the mean of equal waits may exceed their maximum by a rounding error)
    assert 0 < tl["step"] < 1e4 and 0 <= tl["wait_for_peer_flags"] <= tl["longest_flag_wait"] * (1 + 1e-9) < 1e3 and tl["local_slab_reduce"] > 0
    with pytest.raises(RuntimeError, match="peer-to-peer"):
        ref.profile_sharded_steps(2, 0)
    with pytest.raises(RuntimeError, match="cannot be switched off"):
        e.set_p2p(False)


# ------------------------------------------------------------------ MvNMF steps queued ahead of the host on sharded engines (round 5)
MV_CASES = {
    # problems whose line search backtracks at several step indices (found with the oracle: tests/golden/make_golden.py's search,
    # larger cohorts): gammas 1, 1, .96, .74, .71, .68, .65, .78, .75, .72 and 1, 1, 1, 1, .61, .38, .23, .18, .14, .11
    "mixed": dict(N=330, K=3, lam=2163.8587982671297, delta=0.18254529653645993, mean=26.27523189893783, seed=2003),
    "late": dict(N=150, K=8, lam=2533.9897774426486, delta=0.05722437618654012, mean=11.53329221488821, seed=2002),
}
MV_STEPS = 10


def _mv_problem(case):
    from oracle import klnmf_oracle as orc

    c = MV_CASES[case]
    X, W0, H0 = orc.synthetic_problem(V, c["N"], c["K"], seed=c["seed"], mean_mutations=c["mean"])
    return X.clip(orc.EPSILON), W0, H0, c


def _mv_worker(rank, world, port, out_dir, case):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from salamander_amd.distributed import attach_peer_exchange, shard_bounds
    from salamander_amd.engine import Engine

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, W0, H0, c = _mv_problem(case)
        a, b = shard_bounds(c["N"], world, rank)
        e = Engine(b - a, V, c["K"])
        e.upload_X(X[a:b])
        attach_peer_exchange(e)
        out = {}
        for queued in (True, False):
            e.set_mv_queued(queued)
            for mode in ("one", "blocks", "single"):
                e.upload_W(W0), e.upload_H(H0[a:b])
                gamma, gs, fs = 1.0, [], []
                if mode == "one":
                    gamma, f = e.mv_step_objective(MV_STEPS, 0, c["lam"], c["delta"], gamma)
                    gs, fs = [gamma], [f]
                elif mode == "blocks":
                    left = MV_STEPS
                    while left > 0:
                        n = min(3, left)
                        gamma, f = e.mv_step_objective(n, 0, c["lam"], c["delta"], gamma, more_follows=left > n)
                        gs.append(gamma), fs.append(f)
                        left -= n
                else:
                    for _ in range(MV_STEPS):
                        gamma = e.mv_step(1, 0, c["lam"], c["delta"], gamma)
                        gs.append(gamma)
                tag = f"{'q' if queued else 'c'}_{mode}"
                out[tag + "_W"], out[tag + "_H"], out[tag + "_g"], out[tag + "_f"] = e.download_W(), e.download_H(), np.array(gs), np.array(fs)
                out[tag + "_obj"] = e.mv_objective(c["lam"], c["delta"])
        np.savez(os.path.join(out_dir, f"mv{rank}.npz"), a=a, b=b, **out)
        dist.barrier()
        e.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "mixed"), (3, "mixed"), (2, "late"), (3, "late")])
def test_sharded_mvnmf_steps_queued_ahead_of_the_host_equal_the_classic_form(tmp_path, world, case):
    """Round 5: sample-sharded engines take the queued form too -- per step a tail, ONE exchange of
    [G | rowsums_H | KL | the previous trial's KL], the root / trial kernel with the device-side line-search decision, and one
    pass over the samples -- instead of two passes and a host decision per step.  Ranks as processes on one GPU, peer-to-peer
    exchange: bit for bit the classic sharded form (all steps in one call, blocks of three that leave the engines ahead,
    step by step; first trials rejected in the middle of a queue, as its last step and as its first), W identical on all
    ranks, and the unsharded oracle's gamma sequence."""
    from oracle import klnmf_oracle as orc

    mp.spawn(_mv_worker, args=(world, _free_port(), str(tmp_path), case), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"mv{r}.npz")) for r in range(world)]
    X, W0, H0, c = _mv_problem(case)
    W, H, g, gs = W0.T, H0.T, 1.0, []
    for _ in range(MV_STEPS):
        W, H, g = orc.mvnmf_step(X.T, W, H, c["lam"], c["delta"], g, 0)
        gs.append(g)
    assert min(gs) < 1.0 and max(gs[1:]) >= gs[1]  # (the case does reject first trials)
    for mode in ("one", "blocks", "single"):
        for p in parts:
            for key in ("W", "H", "g", "f", "obj"):
                assert np.array_equal(p[f"q_{mode}_{key}"], p[f"c_{mode}_{key}"]), (mode, key)
            assert np.array_equal(p[f"q_{mode}_W"], parts[0][f"q_{mode}_W"])  # replicated W: the same bits on every rank
            assert np.array_equal(p[f"q_{mode}_W"], parts[0]["q_one_W"]) and np.array_equal(p[f"q_{mode}_H"], p["q_one_H"])
        got = parts[0][f"q_{mode}_g"]
        want = {"one": gs[-1:], "blocks": [gs[2], gs[5], gs[8], gs[9]], "single": gs}[mode]
        assert np.allclose(got, want, rtol=1e-12), (mode, got, want)
    assert rel_l2(parts[0]["q_one_W"], W.T) < 1e-7
    assert rel_l2(np.concatenate([p["q_one_H"] for p in parts], axis=0), H.T) < 1e-7
    assert np.isclose(float(parts[0]["q_one_obj"]), orc.kl_divergence_penalized(X.T, W, H, c["lam"], c["delta"]), rtol=1e-9)


# ---- sample shards of engines with feature blocks (> 96 features) or signature chunks (> 64 signatures): the numerators of
# all blocks / chunks cross the ranks in ONE exchange per W update (salnmf.hip: blocked_numerators, chunk_passes)
SPLIT_CASES = {
    "blocks": dict(V=288, K=20, N=1700, n_given=0),      # three feature blocks
    "ragged_blocks": dict(V=250, K=12, N=1500, n_given=3),  # last block 58 wide, given signatures
    "chunks": dict(V=96, K=100, N=1700, n_given=0),      # two signature chunks (64 + 36)
    "chunks_given": dict(V=96, K=130, N=1300, n_given=70),  # the first chunk all given: its numerators are never formed
    "blocks_and_chunks": dict(V=192, K=70, N=1500, n_given=2),  # two feature blocks x two signature chunks
}
SPLIT_STEPS = 6
SPLIT_LAM, SPLIT_DELTA = 0.7, 0.9


def _split_problem(case):
    from oracle import klnmf_oracle as orc

    c = SPLIT_CASES[case]
    X, W0, H0 = orc.synthetic_problem(c["V"], c["N"], c["K"], seed=11)
    wkl = np.random.default_rng(3).uniform(0.5, 2.0, c["N"])
    return c, X, W0, H0, wkl


def _corr_state(c):
    rng = np.random.default_rng(17)
    return rng.normal(0, 0.3, c["K"]), rng.normal(0, 0.4, (c["K"], 6)), rng.normal(0, 0.4, (c["N"], 6))


def _lib_const(name):
    from salamander_amd import _lib

    return getattr(_lib, name)


def _split_worker(rank, world, port, out_dir, case):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from salamander_amd.distributed import attach_peer_exchange, shard_bounds
    from salamander_amd.engine import Engine

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c, X, W0, H0, wkl = _split_problem(case)
        a, b = shard_bounds(c["N"], world, rank)
        e = Engine(b - a, c["V"], c["K"])
        e.upload_X(X[a:b]), e.upload_W(W0), e.upload_H(H0[a:b])
        attach_peer_exchange(e)
        assert e.comm_info() == (world, rank, c["N"])
        e.kl_step(SPLIT_STEPS, c["n_given"])
        Wk, Hk, obj = e.download_W(), e.download_H(), e.objective()
        # weighted steps, then the W half alone with the other clip rule (update_W, _utils_klnmf.py:164-217)
        e.set_weights(wkl[a:b], None)
        e.kl_step(2, c["n_given"])
        e.update_W(c["n_given"])
        Ww, Hw, objw = e.download_W(), e.download_H(), e.objective()
        extra = {}
        if c["K"] <= 64:
            # one CorrNMF update's signature side on the shards (corrnmf_det.py:64-85): aux + numerators of every feature
            # block, one exchange, the W update; the Poisson log-likelihood and the scalings' sums all-reduced
            e.set_weights(None, None)
            e.upload_W(W0)
            beta, L, U = _corr_state(c)
            e.corr_configure(U.shape[1])
            e.corr_upload(_lib_const("CORR_SIGNATURE_SCALINGS"), beta)
            e.corr_upload(_lib_const("CORR_SIGNATURE_EMBEDDINGS"), L)
            e.corr_upload(_lib_const("CORR_SAMPLE_EMBEDDINGS"), U[a:b])
            e.corr_update_sample_scalings(), e.corr_compute_exposures(), e.corr_compute_aux()
            e.corr_update_signatures(c["n_given"]), e.corr_update_signature_scalings()
            extra = dict(Wc=e.download_W(), betac=e.corr_download(_lib_const("CORR_SIGNATURE_SCALINGS")), llh=e.corr_poisson_llh())
        # MvNMF steps on the shards (the plain form of the split engines: numerators and rowsums_H all-reduced, the line
        # search's objectives all-reduced scalars -- every rank takes the same decisions)
        e.set_weights(None, None)
        e.upload_W(W0), e.upload_H(H0[a:b])
        gamma, mv_obj = e.mv_step_objective(3, c["n_given"], SPLIT_LAM, SPLIT_DELTA, 1.0)
        extra.update(Wm=e.download_W(), Hm=e.download_H(), gamma=gamma, mv_obj=mv_obj)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), Wk=Wk, Hk=Hk, obj=obj, Ww=Ww, Hw=Hw, objw=objw, **extra)
        dist.barrier()  # nobody frees its inbox while a peer may still be inside an exchange
        e.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,case", [(2, "blocks"), (3, "ragged_blocks"), (2, "chunks"), (3, "chunks_given"), (2, "blocks_and_chunks")])
def test_sharded_steps_of_engines_with_feature_blocks_or_signature_chunks(tmp_path, world, case):
    from oracle import klnmf_oracle as orc

    mp.spawn(_split_worker, args=(world, _free_port(), str(tmp_path), case), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    c, X, W0, H0, wkl = _split_problem(case)
    ng = c["n_given"]
    W, H = W0.T, H0.T
    for _ in range(SPLIT_STEPS):
        W, H = orc.update_WH(X.T, W, H, None, None, ng)
    for p in parts[1:]:
        assert np.array_equal(parts[0]["Wk"], p["Wk"]) and np.array_equal(parts[0]["Ww"], p["Ww"])  # the same bits on every rank
        assert p["obj"] == parts[0]["obj"] and p["objw"] == parts[0]["objw"]
    assert np.array_equal(parts[0]["Wk"][:ng], W0[:ng])
    assert rel_l2(parts[0]["Wk"], W.T) < 1e-11
    assert rel_l2(np.concatenate([p["Hk"] for p in parts], axis=0), H.T) < 1e-11
    assert np.isclose(float(parts[0]["obj"]), orc.kl_divergence(X.T, W, H), rtol=1e-10)
    for _ in range(2):
        W, H = orc.update_WH(X.T, W, H, wkl, None, ng)
    W = orc.update_W(X.T, W, H, wkl, ng)
    assert rel_l2(parts[0]["Ww"], W.T) < 1e-11
    assert rel_l2(np.concatenate([p["Hw"] for p in parts], axis=0), H.T) < 1e-11
    assert np.isclose(float(parts[0]["objw"]), orc.kl_divergence(X.T, W, H, wkl), rtol=1e-10)
    if c["K"] <= 64:
        # the CorrNMF signature side: equal to one engine that holds all samples (sums over the samples in another order)
        from salamander_amd.engine import Engine

        beta, L, U = _corr_state(c)
        e1 = Engine(c["N"], c["V"], c["K"])
        e1.upload_X(X), e1.upload_W(W0)
        e1.corr_configure(U.shape[1])
        e1.corr_upload(_lib_const("CORR_SIGNATURE_SCALINGS"), beta)
        e1.corr_upload(_lib_const("CORR_SIGNATURE_EMBEDDINGS"), L)
        e1.corr_upload(_lib_const("CORR_SAMPLE_EMBEDDINGS"), U)
        e1.corr_update_sample_scalings(), e1.corr_compute_exposures(), e1.corr_compute_aux()
        e1.corr_update_signatures(ng), e1.corr_update_signature_scalings()
        for p in parts[1:]:
            assert np.array_equal(parts[0]["Wc"], p["Wc"]) and np.array_equal(parts[0]["betac"], p["betac"]) and p["llh"] == parts[0]["llh"]
        assert rel_l2(parts[0]["Wc"], e1.download_W()) < 1e-12
        assert rel_l2(parts[0]["betac"], e1.corr_download(_lib_const("CORR_SIGNATURE_SCALINGS"))) < 1e-12
        assert np.isclose(float(parts[0]["llh"]), e1.corr_poisson_llh(), rtol=1e-12)
        e1.close()
    W, H, g = W0.T, H0.T, 1.0
    for _ in range(3):
        W, H, g = orc.mvnmf_step(X.T, W, H, SPLIT_LAM, SPLIT_DELTA, g, ng)
    for p in parts[1:]:
        assert np.array_equal(parts[0]["Wm"], p["Wm"]) and p["gamma"] == parts[0]["gamma"] and p["mv_obj"] == parts[0]["mv_obj"]
    assert np.isclose(float(parts[0]["gamma"]), g, rtol=1e-12)
    assert rel_l2(parts[0]["Wm"], W.T) < 1e-7
    assert rel_l2(np.concatenate([p["Hm"] for p in parts], axis=0), H.T) < 1e-7
    assert np.isclose(float(parts[0]["mv_obj"]), orc.kl_divergence_penalized(X.T, W, H, SPLIT_LAM, SPLIT_DELTA), rtol=1e-9)


# ---- a whole CorrNMF update on sample shards, ranks as processes on one GPU (round 5): the first run of the sharded CorrNMF
# path with more than one rank on a GPU.  No RCCL (two RCCL ranks cannot share a device): every sum over the samples --
# numerators, scalings, the lockstep signature solves' evaluation records (66 + dim^2 doubles per signature: they fit the
# peer inbox at this dim), the embedding norms, the Poisson log-likelihood -- goes through the peer exchange.
CORR_SHARD = dict(V=96, K=12, N=9000, dim=6, variance=0.8)


def _corr_shard_problem():
    from oracle import klnmf_oracle as orc

    c = CORR_SHARD
    X, W, _ = orc.synthetic_problem(c["V"], c["N"], c["K"], seed=21)
    rng = np.random.default_rng(23)
    return c, X, W, rng.normal(0, 0.3, c["K"]), rng.normal(0, 0.4, (c["K"], c["dim"])), rng.normal(0, 0.4, (c["N"], c["dim"]))


def _corr_update(e, c, n_updates=2):
    from salamander_amd import _lib

    status = None
    for _ in range(n_updates):  # (corrnmf_det.py:64-141: the order of CorrNMFDet._update_parameters)
        e.corr_update_sample_scalings(), e.corr_compute_exposures(), e.corr_compute_aux()
        e.corr_update_signatures(0), e.corr_update_signature_scalings()
        status = e.corr_update_signature_embeddings(c["variance"], 0, return_status=True)
        e.corr_update_sample_embeddings(c["variance"], 3)
    sumsq, llh = e.corr_embedding_sumsq(), e.corr_poisson_llh()
    return dict(W=e.download_W(), beta=e.corr_download(_lib.CORR_SIGNATURE_SCALINGS), L=e.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS),
                U=e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS), alpha=e.corr_download(_lib.CORR_SAMPLE_SCALINGS), status=status,
                sumsq=np.array(sumsq), llh=llh)


def _corr_shard_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    from salamander_amd import _lib
    from salamander_amd.distributed import attach_peer_exchange, shard_bounds
    from salamander_amd.engine import Engine

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c, X, W, beta, L, U = _corr_shard_problem()
        a, b = shard_bounds(c["N"], world, rank)
        e = Engine(b - a, c["V"], c["K"])
        e.upload_X(X[a:b]), e.upload_W(W)
        attach_peer_exchange(e)
        e.corr_configure(c["dim"])
        e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, beta), e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L)
        e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U[a:b])
        out = _corr_update(e, c)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), a=a, b=b, **out)
        dist.barrier()  # nobody frees its inbox while a peer may still be inside an exchange
        e.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_corrnmf_update_over_the_peer_exchange_matches_one_engine(tmp_path, world):
    from salamander_amd import _lib
    from salamander_amd.engine import Engine

    mp.spawn(_corr_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(os.path.join(tmp_path, f"rank{r}.npz")) for r in range(world)]
    c, X, W, beta, L, U = _corr_shard_problem()
    e1 = Engine(c["N"], c["V"], c["K"])
    e1.upload_X(X), e1.upload_W(W)
    e1.corr_configure(c["dim"])
    e1.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, beta), e1.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L)
    e1.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U)
    want = _corr_update(e1, c)
    e1.close()
    for p in parts[1:]:  # the replicated state: the same bits on every rank (sums in rank order)
        for key in ("W", "beta", "L", "sumsq"):
            assert np.array_equal(parts[0][key], p[key]), key
        assert p["llh"] == parts[0]["llh"] and np.array_equal(p["status"], parts[0]["status"])
    assert np.array_equal(parts[0]["status"], want["status"])
    # against the engine that holds all samples: the sums over the samples in another order -- rounding level on the dense
    # pieces, a few digits more through two updates' Newton-CG solves
    assert rel_l2(parts[0]["W"], want["W"]) < 1e-11 and rel_l2(parts[0]["beta"], want["beta"]) < 1e-9
    assert rel_l2(parts[0]["L"], want["L"]) < 1e-7
    assert rel_l2(np.concatenate([p["U"] for p in parts]), want["U"]) < 1e-7
    assert rel_l2(np.concatenate([p["alpha"] for p in parts]), want["alpha"]) < 1e-9
    assert np.allclose(parts[0]["sumsq"], want["sumsq"], rtol=1e-7) and np.isclose(float(parts[0]["llh"]), want["llh"], rtol=1e-9)
