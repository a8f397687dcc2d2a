"""HIP path vs oracle / golden vectors through the C ABI, on an MI355X.

Tolerance: the north star asks for W, H within 1e-4 rel-L2 of the reference CPU path after
equal iteration counts.  The engine computes in fp64 with the reference's operation order
inside each element (only GEMM summation order differs), so the tests assert the much
tighter ``TOL`` below; the 1e-4 contract is therefore met with eight orders of margin.
"""

import numpy as np
import pytest

import salamander_amd as sal
from conftest import rel_l2
from oracle import klnmf_oracle as orc
from salamander_amd import Engine, _lib
from salamander_amd.models import _utils_klnmf

pytestmark = pytest.mark.gpu

TOL = 1e-10       # rel-L2 on W, H after up to 100 steps
TOL_LONG = 1e-8   # after 500 steps
EPS = orc.EPSILON


def make_engine(X, W, H, wkl=None, wlh=None):
    """X (V,N), W (V,K), H (K,N) in reference shapes -> engine holding the sample-major copies."""
    e = Engine(X.shape[1], X.shape[0], W.shape[1])
    e.upload_X(X.T)
    e.upload_W(W.T)
    e.upload_H(H.T)
    e.set_weights(wkl, wlh)
    return e


# ------------------------------------------------------------------ golden trajectories (generated from the reference)
@pytest.mark.parametrize(
    "tag,use_wkl,use_wlh,n_given,marks",
    [
        ("plain", False, False, 0, (1, 10, 100)),
        ("wkl", True, False, 0, (1, 10)),
        ("lhalf", True, True, 0, (1, 10)),
        ("given7", False, False, 7, (1, 10)),
        ("given50", False, True, 50, (1, 10)),
    ],
)
def test_golden_trajectories(golden, tag, use_wkl, use_wlh, n_given, marks):
    g = golden.synth
    wk = g["wkl"] if use_wkl else None
    wl = g["wlh"] if use_wlh else None
    e = make_engine(g["X"], g["W0"], g["H0"], wk, wl)
    objs, t = [e.objective()], 0
    for mark in sorted(set(marks) | set(range(10, max(marks) + 1, 10))):
        e.kl_step(mark - t, n_given)
        t = mark
        if t in marks:
            assert rel_l2(e.download_W(), g[f"{tag}_W{t}"].T) < TOL
            assert rel_l2(e.download_H(), g[f"{tag}_H{t}"].T) < TOL
        if t % 10 == 0:
            objs.append(e.objective())
    assert np.allclose(objs, g[f"{tag}_obj"], rtol=1e-11)
    if n_given:
        assert np.array_equal(e.download_W()[:n_given], g["W0"].T[:n_given].clip(EPS) if n_given < 50 else g["W0"].T)
    e.close()


def test_golden_pcawg_c1_engine(golden):
    """Config c1 data (data/pcawg_breast_sbs.csv, K=5): 500 steps against the reference-generated trajectory."""
    g = golden.pcawg
    e = make_engine(g["X"], g["W0"], g["H0"])
    t, objs = 0, []
    for mark in range(10, 501, 10):
        e.kl_step(mark - t)
        t = mark
        objs.append(e.objective())
        if t in (10, 100, 500):
            assert rel_l2(e.download_W(), g[f"W{t}"].T) < TOL and rel_l2(e.download_H(), g[f"H{t}"].T) < TOL
    assert np.allclose(objs, g["obj"][1:], rtol=1e-12)
    e.close()


def test_pcawg_c1_model_fit(golden):
    """Config c1 through the model API: KLNMF(n_signatures=5).fit(adata) with a custom init.

    ``fit`` post-processes the custom init exactly as the reference does (normalise, clip:
    initialize.py:116-118), so the expected trajectory is the oracle's fit from that init."""
    from salamander_amd.initialization import initialize_mat

    g = golden.pcawg
    adata = sal.AnnData(g["X"].T.copy())
    m = sal.models.KLNMF(5, "custom", min_iterations=500, max_iterations=500)
    m.fit(adata, init_kwargs={"signatures_mat": g["W0"].T.copy(), "exposures_mat": g["H0"].T.copy()})
    S0, E0 = initialize_mat(g["X"].T, 5, "custom", signatures_mat=g["W0"].T.copy(), exposures_mat=g["H0"].T.copy())
    W, H, it, hist = orc.fit_klnmf(g["X"], S0.T, E0.T, min_iterations=500, max_iterations=500)
    assert m.n_iterations_ == it == 500
    assert np.allclose(m.history["objective_function"], hist, rtol=1e-11)
    assert rel_l2(m.asignatures.X, W.T) < TOL_LONG and rel_l2(adata.obsm["exposures"], H.T) < TOL_LONG


def test_golden_objectives_with_zeros(golden):
    """X == 0 branches of kl_divergence (:47-50) and samplewise_kl_divergence (:82-86)."""
    g = golden.synth
    assert np.allclose(_utils_klnmf.kl_divergence(g["Xz"], g["W0"], g["H0"]), g["kl_z"], rtol=1e-12)
    assert np.allclose(_utils_klnmf.samplewise_kl_divergence(g["Xz"], g["W0"], g["H0"]), g["skl_z"], rtol=1e-11)
    assert np.allclose(_utils_klnmf.samplewise_kl_divergence(g["X"], g["W0"], g["H0"]), g["skl0"], rtol=1e-11)
    assert np.allclose(_utils_klnmf.update_H(g["X"], g["W0"], g["H0"]), g["H_upd"], rtol=1e-11)
    assert np.allclose(_utils_klnmf.update_W(g["X"], g["W0"], g["H0"], g["wkl"], 7), g["W_upd_g7"], rtol=1e-11)


def test_golden_small_every_function(golden):
    import os

    from conftest import REF_FIX

    g = golden.small
    X, wkl, wlh = g["X"], g["wkl"], g["wlh"]
    for K in (1, 2):
        t = f"k{K}_"
        W = np.load(os.path.join(REF_FIX, "utils_klnmf", f"W_nsigs{K}.npy"))
        H = np.load(os.path.join(REF_FIX, "utils_klnmf", f"H_nsigs{K}.npy"))
        tight = dict(rtol=1e-11, atol=1e-14)
        assert np.allclose(_utils_klnmf.kl_divergence(X, W, H, wkl), g[t + "kl_w"], **tight)
        assert np.allclose(_utils_klnmf.samplewise_kl_divergence(X, W, H, wkl), g[t + "skl_w"], **tight)
        assert np.allclose(_utils_klnmf.update_W(X, W, H, wkl), g[t + "W_upd_w"], **tight)
        assert np.allclose(_utils_klnmf.update_W(X, W, H, None, 1), g[t + "W_upd_g1"], **tight)
        assert np.allclose(_utils_klnmf.update_H(X, W, H, wkl, wlh), g[t + "H_upd_lh"], **tight)
        assert np.allclose(_utils_klnmf.update_H(X, W, H, None, wlh), g[t + "H_upd_lh_nokl"], **tight)
        Wj, Hj = _utils_klnmf.update_WH(X, W, H, wkl, wlh, 1)
        assert np.allclose(Wj, g[t + "Wj_all"], **tight) and np.allclose(Hj, g[t + "Hj_all"], **tight)


# ------------------------------------------------------------------ MvNMF incl. the backtracking branch
@pytest.mark.parametrize("tag", ["a", "g", "bt1", "bt2"])
def test_golden_mvnmf(golden, tag):
    g = golden.mv
    lam, delta, steps, ng = g[f"{tag}_par"]
    e = make_engine(g[f"{tag}_X"], g[f"{tag}_W0"], g[f"{tag}_H0"])
    assert np.isclose(e.mv_objective(lam, delta), g[f"{tag}_obj0"], rtol=1e-11)
    gamma, gammas = 1.0, []
    for _ in range(int(steps)):
        gamma = e.mv_step(1, int(ng), lam, delta, gamma)
        gammas.append(gamma)
    assert np.allclose(gammas, g[f"{tag}_gammas"], rtol=1e-12), (gammas, g[f"{tag}_gammas"])
    assert rel_l2(e.download_W(), g[f"{tag}_W"].T) < 1e-7 and rel_l2(e.download_H(), g[f"{tag}_H"].T) < 1e-7
    assert np.isclose(e.mv_objective(lam, delta), g[f"{tag}_obj"], rtol=1e-9)
    e.close()


@pytest.mark.parametrize("tag", ["a", "g", "bt1", "bt2"])
def test_golden_mvnmf_all_steps_in_one_call(golden, tag):
    """The same trajectories with all steps in ONE mv_step call: inside a call the next step's update_H pass and
    W-only algebra run speculatively while the host waits for the line-search scalars; the backtracking cases
    (bt1, bt2) make that speculation fail and be dropped.  Then a lazily rescaled H must survive other readers."""
    g = golden.mv
    lam, delta, steps, ng = g[f"{tag}_par"]
    e = make_engine(g[f"{tag}_X"], g[f"{tag}_W0"], g[f"{tag}_H0"])
    gamma = e.mv_step(int(steps), int(ng), lam, delta, 1.0)
    assert np.isclose(gamma, g[f"{tag}_gammas"][-1], rtol=1e-12)
    assert np.isclose(e.mv_objective(lam, delta), g[f"{tag}_obj"], rtol=1e-9)  # reads H through the pending rescale
    skl = e.samplewise_kl()
    assert rel_l2(e.download_W(), g[f"{tag}_W"].T) < 1e-7 and rel_l2(e.download_H(), g[f"{tag}_H"].T) < 1e-7
    assert np.allclose(skl, orc.samplewise_kl_divergence(g[f"{tag}_X"], g[f"{tag}_W"], g[f"{tag}_H"]), rtol=1e-6)
    # split differently: 3 steps + the rest, with a KL-NMF update_W in between left out -- same result as one call
    e2 = make_engine(g[f"{tag}_X"], g[f"{tag}_W0"], g[f"{tag}_H0"])
    k = min(3, int(steps) - 1)
    gamma2 = e2.mv_step(k, int(ng), lam, delta, 1.0)
    gamma2 = e2.mv_step(int(steps) - k, int(ng), lam, delta, gamma2)
    assert gamma2 == gamma and np.array_equal(e2.download_W(), e.download_W()) and np.array_equal(e2.download_H(), e.download_H())
    e.close()
    e2.close()


@pytest.mark.parametrize("tag", ["a", "g", "bt1", "bt2"])
def test_mv_step_returns_the_objective_of_the_state_it_leaves(golden, tag):
    """``mv_step_objective``: the last line search's accepted value IS ``MvNMF.objective_function`` of the new state
    (mvnmf.py:82-89 vs :149-156) -- equal to ``mv_objective`` to rounding, through accepted first trials and backtracking
    (bt1, bt2), in one call and step by step, same state bit for bit as ``mv_step``; all signatures given: the objective as
    a pass of its own."""
    g = golden.mv
    lam, delta, steps, ng = g[f"{tag}_par"]
    a, b = (make_engine(g[f"{tag}_X"], g[f"{tag}_W0"], g[f"{tag}_H0"]) for _ in range(2))
    ga = a.mv_step(int(steps), int(ng), lam, delta, 1.0)
    gb, fb = b.mv_step_objective(int(steps), int(ng), lam, delta, 1.0)
    assert ga == gb and np.array_equal(a.download_W(), b.download_W()) and np.array_equal(a.download_H(), b.download_H())
    assert np.isclose(fb, a.mv_objective(lam, delta), rtol=1e-12, atol=0) and np.isclose(fb, g[f"{tag}_obj"], rtol=1e-9)
    gamma = gb
    for _ in range(3):
        gamma, f = b.mv_step_objective(1, int(ng), lam, delta, gamma)
        assert np.isclose(f, b.mv_objective(lam, delta), rtol=1e-12, atol=0)
    K = b.K
    _, f = b.mv_step_objective(2, K, lam, delta, gamma)
    assert f == b.mv_objective(lam, delta)
    a.close(), b.close()


@pytest.mark.parametrize("tag", ["a", "g", "bt1", "bt2"])
def test_mv_steps_left_ahead_between_calls(golden, tag):
    """``mv_step_objective(more_follows=True)`` keeps the speculative first half of the next step across calls: blocks of
    steps chained that way, with and without other calls in between (which make the engine step back: objective, download,
    a changed delta), end in the same state bit for bit as the plain calls, with the same gammas and objectives."""
    g = golden.mv
    lam, delta, steps, ng = g[f"{tag}_par"]
    steps, ng = int(steps), int(ng)
    a, b, c = (make_engine(g[f"{tag}_X"], g[f"{tag}_W0"], g[f"{tag}_H0"]) for _ in range(3))
    ga, gb, gc = 1.0, 1.0, 1.0
    fa, fb = [], []
    done = 0
    while done < steps:
        n = min(2, steps - done)
        ga, f = a.mv_step_objective(n, ng, lam, delta, ga)
        fa.append(f)
        gb, f = b.mv_step_objective(n, ng, lam, delta, gb, more_follows=True)
        fb.append(f)
        gc, f = c.mv_step_objective(n, ng, lam, delta, gc, more_follows=True)
        # (c steps back for the call in between, then resumes afresh; the value of a speculative last step comes from the
        # update_H pass's in-launch sum instead of a forward pass: equal to rounding)
        assert np.isclose(f, fa[-1], rtol=1e-12, atol=0) and np.isclose(c.mv_objective(lam, delta), f, rtol=1e-12, atol=0)
        done += n
    assert ga == gb == gc and np.allclose(fa, fb, rtol=1e-12, atol=0)
    Wa, Ha = a.download_W(), a.download_H()
    assert np.array_equal(b.download_W(), Wa) and np.array_equal(b.download_H(), Ha)
    assert np.array_equal(c.download_W(), Wa) and np.array_equal(c.download_H(), Ha)
    # left ahead, then a different continuation: KL steps, and an MvNMF step with another delta
    ga, _ = a.mv_step_objective(1, ng, lam, delta, ga)
    gb, _ = b.mv_step_objective(1, ng, lam, delta, gb, more_follows=True)
    a.kl_step(2, ng), b.kl_step(2, ng)
    ga, _ = a.mv_step_objective(1, ng, lam, delta, ga)
    gb, _ = b.mv_step_objective(1, ng, lam, delta, gb, more_follows=True)
    ga, f1 = a.mv_step_objective(1, ng, lam, 2.0 * delta, ga)
    gb, f2 = b.mv_step_objective(1, ng, lam, 2.0 * delta, gb, more_follows=True)
    assert ga == gb and np.isclose(f1, f2, rtol=1e-12, atol=0)
    assert np.array_equal(b.download_W(), a.download_W()) and np.array_equal(b.download_H(), a.download_H())
    a.close(), b.close(), c.close()


# ------------------------------------------------------------------ shapes: ragged N, V < 96, every K bucket
@pytest.mark.parametrize(
    "V,N,K",
    [(96, 1, 1), (96, 15, 3), (96, 16, 4), (96, 17, 5), (96, 63, 8), (96, 64, 9), (96, 250, 16), (96, 1000, 17),
     (96, 999, 30), (96, 333, 33), (96, 1025, 40), (96, 777, 41), (96, 4099, 50), (96, 300, 52), (96, 301, 53),
     (96, 500, 64), (83, 777, 30), (83, 200, 40), (7, 100, 2), (16, 40, 5),
     # more tiles than waves with a short leftover round: the cooperative kernel runs
     # (leftover tile worked on by the four waves of a workgroup, every output-side geometry: KT = 1..4, KR = 0..4)
     (96, 17190, 50), (96, 17001, 30), (83, 17100, 34), (96, 16500, 64), (96, 16390, 20),
     (96, 16450, 1), (96, 16999, 5), (7, 17000, 16), (96, 17010, 17), (96, 16800, 36), (96, 17111, 48), (96, 16777, 53),
     (96, 16601, 19), (96, 33500, 51)],
)
def test_shapes_one_and_five_steps(V, N, K):
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=V + N + K)
    rng = np.random.default_rng(1)
    wkl, wlh = rng.uniform(0.5, 2, N), rng.uniform(0, 3, N)
    for wk, wl, ng in ((None, None, 0), (wkl, wlh, min(2, K))):
        e = Engine(N, V, K)
        e.upload_X(X)
        e.upload_W(W0)
        e.upload_H(H0)
        e.set_weights(wk, wl)
        assert np.isclose(e.objective(), orc.klnmf_objective(X.T, W0.T, H0.T, wk, wl), rtol=1e-12)
        W, H = W0.T, H0.T
        for steps in (1, 4):
            for _ in range(steps):
                W, H = orc.update_WH(X.T, W, H, wk, wl, ng)
            e.kl_step(steps, ng)
            assert rel_l2(e.download_W(), W.T) < TOL and rel_l2(e.download_H(), H.T) < TOL
        assert np.allclose(e.samplewise_kl(), orc.samplewise_kl_divergence(X.T, W, H), rtol=1e-10)
        assert np.allclose(e.reconstruct(), H.T @ W.T, rtol=1e-13)
        e.close()


@pytest.mark.parametrize("K", list(range(1, 65)))
def test_every_signature_count(K):
    """Every K in 1..64 (each (KS, KTM, KR) kernel geometry, incl. all remainder-column variants):
    joint step with weights / l-half / given signatures, update_W, objective, MvNMF step."""
    V, N = 96, 203 + K  # ragged last tile
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=K)
    rng = np.random.default_rng(K)
    wkl, wlh = rng.uniform(0.5, 2, N), rng.uniform(0, 3, N)
    ng = K // 3
    e = Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.set_weights(wkl, wlh)
    W, H = W0.T, H0.T
    for _ in range(2):
        W, H = orc.update_WH(X.T, W, H, wkl, wlh, ng)
    e.kl_step(2, ng)
    assert rel_l2(e.download_W(), W.T) < TOL and rel_l2(e.download_H(), H.T) < TOL
    assert np.isclose(e.objective(), orc.klnmf_objective(X.T, W, H, wkl, wlh), rtol=1e-12)
    e.update_W(ng, _lib.CLIP_NON_GIVEN)
    assert rel_l2(e.download_W(), orc.update_W(X.T, W, H, wkl, ng).T) < TOL
    # MvNMF step from the same state (unweighted)
    e.set_weights(None, None)
    e.upload_W(W.T.copy()), e.upload_H(H.T.copy())
    Wm, Hm, g = orc.mvnmf_step(X.T, W, H, 1.0, 1.0, 1.0, ng)
    gg = e.mv_step(1, ng, 1.0, 1.0, 1.0)
    assert gg == g
    assert rel_l2(e.download_W(), Wm.T) < 1e-7 and rel_l2(e.download_H(), Hm.T) < 1e-7
    e.close()


def test_unnormalised_inputs_and_function_level_semantics():
    """Function-level calls take any non-negative W, H (not only normalised ones)."""
    rng = np.random.default_rng(3)
    V, N, K = 96, 130, 6
    X = rng.poisson(30, size=(V, N)).astype(float)
    W, H = rng.random((V, K)) * 3, rng.random((K, N)) * 5
    Wj, Hj = _utils_klnmf.update_WH(X, W, H, n_given_signatures=2)
    Wo, Ho = orc.update_WH(X, W, H, n_given_signatures=2)
    assert rel_l2(Wj, Wo) < TOL and rel_l2(Hj, Ho) < TOL
    assert rel_l2(_utils_klnmf.update_W(X, W, H, None, 2), orc.update_W(X, W, H, None, 2)) < TOL
    assert rel_l2(_utils_klnmf.update_H(X, W, H), orc.update_H(X, W, H)) < TOL
    assert np.isclose(_utils_klnmf.kl_divergence(X, W, H), orc.kl_divergence(X, W, H), rtol=1e-12)
    # all signatures given: W comes back untouched (not even clipped), H still updates
    Wg, Hg = _utils_klnmf.update_WH(X, W, H, n_given_signatures=K)
    assert np.array_equal(Wg, W) and rel_l2(Hg, Ho) < TOL
    with pytest.raises(ValueError):
        _utils_klnmf.update_WH(X, W[:, :3], H)


def test_errors_are_exceptions_not_aborts():
    with pytest.raises(RuntimeError):
        Engine(10, 3073, 2)
    with pytest.raises(RuntimeError):
        Engine(10, 96, 513)
    with pytest.raises(RuntimeError):
        Engine(0, 96, 2)
    e = Engine(10, 96, 2)
    with pytest.raises(ValueError):
        e.upload_X(np.zeros((10, 95)))
    with pytest.raises(RuntimeError):
        e.kl_step(1, 3)
    e.close()


def test_limits_are_refused_with_a_message_and_closed_engines_raise():
    """The documented limits of this build (include/salnmf.h) and the ADVICE r1 NULL-handle case."""
    for args, text in (((10, 3073, 2), "n_features"), ((10, 96, 513), "n_signatures")):
        with pytest.raises(RuntimeError, match=text):
            Engine(*args)
    e = Engine(10, 96, 2)
    with pytest.raises(RuntimeError, match="dim_embeddings"):
        e.corr_configure(65)
    e.close()
    for call in (lambda: e.upload_W(np.ones((2, 96))), lambda: e.download_W(), lambda: e.kl_step(1), lambda: e.objective()):
        with pytest.raises(RuntimeError, match="closed"):
            call()


def test_nan_propagates_as_in_the_reference():
    """ndarray.clip keeps a NaN (`_utils_klnmf.py:341,347`); so do the engine's clips (fmax would have
    replaced it by EPSILON and let a poisoned fit look converged).  Same NaN pattern as the oracle, NaN objective."""
    X, W0, H0 = orc.synthetic_problem(96, 200, 7, seed=12)
    H0[5, 3] = np.nan
    e = Engine(200, 96, 7)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    assert np.isnan(e.objective())
    e.kl_step(1)
    W, H = e.download_W(), e.download_H()
    with np.errstate(invalid="ignore"):
        Wr, Hr = orc.update_WH(X.T, W0.T, H0.T)
    assert np.array_equal(np.isnan(W), np.isnan(Wr.T)) and np.array_equal(np.isnan(H), np.isnan(Hr.T))
    assert np.isnan(H[5]).all() and np.isfinite(H[np.arange(200) != 5]).all()
    e.close()


def test_upload_clip_matches_setup_adata():
    X, W0, H0 = orc.synthetic_problem(96, 100, 4, seed=9)
    X[X <= EPS] = 0.0
    e = Engine(100, 96, 4)
    e.upload_X(X, clip=True)
    e.upload_W(W0)
    e.upload_H(H0)
    assert np.isclose(e.objective(), orc.kl_divergence(X.T.clip(EPS), W0.T, H0.T), rtol=1e-12)
    e.close()


@pytest.mark.parametrize("dtype", ["float64", "float32", "int32", "int64", "uint16"])
@pytest.mark.parametrize("N,V", [(100003, 96), (4000, 83)])
def test_typed_chunked_ingest_equals_host_conversion(dtype, N, V):
    """Row f4: raw count matrices of any supported element type cross PCIe as they are (several 32 MB chunks at
    N = 100 003, ragged last tile, V < 96) and are converted, clipped and padded on the device -- the same bits as
    converting and clipping on the host, as SignatureNMF._setup_adata does (signature_nmf.py:269-281)."""
    rng = np.random.default_rng(N + V)
    counts = rng.poisson(rng.gamma(0.6, 10.0, size=(N, V)))  # many zeros: the clip matters
    raw = counts.astype(dtype)
    K = 5
    H0 = rng.uniform(0.5, 2.0, size=(N, K))
    W0 = rng.dirichlet(np.ones(V), size=K)
    a, b = Engine(N, V, K), Engine(N, V, K)
    a.upload_X(raw, clip=True)                                   # typed, converted on the device
    b.upload_X(raw.astype(np.float64).clip(EPS), clip=False)     # what the reference hands over
    for e in (a, b):
        e.upload_W(W0), e.upload_H(H0)
        e.kl_step(2)
    assert np.array_equal(a.download_H(), b.download_H()) and np.array_equal(a.download_W(), b.download_W())
    assert a.objective() == b.objective()
    assert np.array_equal(a.samplewise_kl(), b.samplewise_kl())
    a.close(), b.close()


def test_persistent_multi_step_launch_gives_the_same_bits():
    """The opt-in persistent kernel (n steps in one launch, in-kernel hand-offs) against per-step
    launches: bit-identical W, H and objective, incl. given signatures, a grid smaller than the CU count and a step
    count that is split over several launches.  The default library does not carry that kernel (a measured negative):
    there the switch must refuse, and the comparison runs only in builds made with SALNMF_WITH_PERSISTENT=1."""
    if not (_lib.load().salnmf_build_flags() & _lib.BUILD_PERSISTENT):
        e = Engine(500, 96, 5)
        with pytest.raises(RuntimeError, match="persistent"):
            e.set_persistent(True)
        e.set_persistent(False)
        e.close()
        return
    for N, K, n_given, steps in ((20000, 50, 0, 7), (3000, 50, 7, 70), (100003, 30, 0, 3), (500, 5, 0, 4)):
        X, W0, H0 = orc.synthetic_problem(96, N, K, seed=N % 97)
        outs = []
        for persistent in (False, True):
            e = Engine(N, 96, K)
            e.set_persistent(persistent)
            e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
            e.kl_step(steps, n_given)
            outs.append((e.download_W(), e.download_H(), e.objective()))
            e.close()
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]


@pytest.mark.parametrize(
    "N,K,n_given,weights,precision",
    [(20000, 50, 0, False, "f64"), (3000, 50, 7, False, "f64"), (1000, 30, 0, True, "f64"), (500, 5, 5, False, "f64"), (4000, 16, 0, False, "f32")],
)
def test_kept_block_rollback_and_queued_objectives(N, K, n_given, weights, precision):
    """``kl_step_keep`` = ``kl_step`` bit for bit, ``kl_rollback`` returns to the state the kept block started from
    (incl. all signatures given, per-sample weights, the fp32 fast mode), and the queued objectives are the blocking
    ones -- what lets ``fit`` launch the next block before it has read the deciding objective."""
    X, W0, H0 = orc.synthetic_problem(96, N, K, seed=N % 89)
    rng = np.random.default_rng(3)
    wk, wl = (rng.uniform(0.5, 2.0, N), rng.uniform(0.0, 0.3, N)) if weights else (None, None)
    engines = []
    for _ in range(2):
        e = Engine(N, 96, K)
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
        e.set_weights(wk, wl)
        e.set_precision(precision)
        e.kl_step(3, n_given)
        engines.append(e)
    a, b = engines
    W3, H3, o3 = a.download_W(), a.download_H(), a.objective()
    a.kl_step(7, n_given)
    b.objective_async(5)
    b.kl_step_keep(7, n_given)
    b.objective_async(6)
    assert np.array_equal(b.download_W(), a.download_W()) and np.array_equal(b.download_H(), a.download_H())
    b.kl_rollback()
    assert np.array_equal(b.download_W(), W3) and np.array_equal(b.download_H(), H3)
    with pytest.raises(RuntimeError, match="kept"):
        b.kl_rollback()
    b.kl_step_keep(7, n_given)  # again from the restored state: the same seven steps
    b.objective_async(7)
    assert np.array_equal(b.download_W(), a.download_W()) and np.array_equal(b.download_H(), a.download_H())
    vals = b.objective_read(5, 3)
    assert vals[0] == o3 and vals[1] == a.objective() and vals[2] == vals[1]
    a.close(), b.close()


@pytest.mark.parametrize(
    "N,V,K,n_given,weights",
    [(16 * 1074 - 5, 96, 50, 0, False), (20000, 96, 30, 3, False), (3001, 83, 7, 0, False), (100000, 96, 50, 0, False), (5000, 96, 64, 0, False),
     (2000, 96, 52, 0, False), (1000, 96, 12, 12, False), (1500, 96, 20, 0, True), (1200, 288, 12, 0, False)],
)
def test_objective_folded_into_the_following_step(N, V, K, n_given, weights):
    """``kl_step_objective``: the objective of the state the steps start from, evaluated inside the first of them
    (``fused_kernel<..., G, U, STATS>`` + the spare workgroup of its tail) -- the value of ``objective()`` to rounding, the
    steps themselves bit for bit ``kl_step`` (plain, kept and rolled back), incl. sizes with a cooperative leftover tile,
    ragged N, V < 96, every accumulator geometry's extreme; the cases the fold does not cover (all signatures given,
    weights, feature blocks, no step behind the objective) fall back to the forward pass: the same bits."""
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=N % 83)
    rng = np.random.default_rng(5)
    wk, wl = (rng.uniform(0.5, 2.0, N), rng.uniform(0.0, 0.3, N)) if weights else (None, None)
    a, b = Engine(N, V, K), Engine(N, V, K)
    for e in (a, b):
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
        e.set_weights(wk, wl)
        e.kl_step(2, n_given)
    folds = not weights and n_given < K and V <= 96
    o2 = a.objective()
    W2, H2 = a.download_W(), a.download_H()
    a.kl_step(4, n_given)
    b.kl_step_objective(3, 4, n_given, keep=True)
    assert np.array_equal(b.download_W(), a.download_W()) and np.array_equal(b.download_H(), a.download_H())
    b.kl_rollback()
    assert np.array_equal(b.download_W(), W2) and np.array_equal(b.download_H(), H2)
    b.kl_step_objective(4, 4, n_given, keep=False)
    b.kl_step_objective(5, 0, n_given)
    assert np.array_equal(b.download_W(), a.download_W()) and np.array_equal(b.download_H(), a.download_H())
    vals = b.objective_read(3, 3)
    if folds:
        assert np.isclose(vals[0], o2, rtol=1e-12, atol=0) and vals[1] == vals[0]
    else:
        assert vals[0] == o2 and vals[1] == o2
    assert vals[2] == a.objective()
    a.close(), b.close()


@pytest.mark.parametrize("K,n_given,lhalf", [(50, 0, False), (50, 5, True), (30, 0, True), (16, 0, False)])
def test_weighted_steps_through_the_cooperative_leftover_tile(K, n_given, lhalf):
    """Per-sample weights at a size whose leftover round is worked on by whole workgroups (1074 tiles on 256 x 4 waves: 50
    cooperative tiles, incl. the ragged last one): the weighted instantiation honours ``weights_kl`` / ``weights_lhalf``
    there as in the ordinary tiles -- against the oracle."""
    N = 16 * 1074 - 5
    X, W0, H0 = orc.synthetic_problem(96, N, K, seed=K)
    rng = np.random.default_rng(K)
    wk = rng.uniform(0.5, 2.0, N)
    wl = rng.uniform(0.0, 0.3, N) if lhalf else None
    e = Engine(N, 96, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.set_weights(wk, wl)
    W, H = W0.T, H0.T
    for _ in range(3):
        W, H = orc.update_WH(X.T, W, H, wk, wl, n_given)
    e.kl_step(3, n_given)
    assert rel_l2(e.download_W(), W.T) < 1e-12 and rel_l2(e.download_H(), H.T) < 1e-12
    assert np.isclose(e.objective(), orc.klnmf_objective(X.T, W, H, wk, wl), rtol=1e-12)
    e.close()


@pytest.mark.parametrize("N,V,K", [(3000, 96, 50), (777, 83, 7), (1200, 288, 12)])
def test_lazy_exposure_scale_of_the_initialisation(N, V, K):
    """``Engine.set_H_scale`` (normalize_WH's exposure side + clip, initialize.py:116-118, applied by the first pass that
    rewrites H): every reader sees clip(H * scale) bit for bit, and the steps continue from exactly that state."""
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=N)
    scale = np.random.default_rng(1).uniform(0.2, 3.0, K)
    Hs = np.clip(H0 * scale[None, :], orc.EPSILON, None)
    a, b = Engine(N, V, K), Engine(N, V, K)
    for e, H in ((a, H0), (b, Hs)):
        e.upload_X(X), e.upload_W(W0), e.upload_H(H)
    a.set_H_scale(scale)
    assert a.objective() == b.objective()
    a.kl_step(3), b.kl_step(3)
    # (to rounding: a pass that applies the scale on the fly does not use the cooperative leftover tile, which changes
    # the order in which the numerator is summed)
    assert rel_l2(a.download_W(), b.download_W()) < 1e-13 and rel_l2(a.download_H(), b.download_H()) < 1e-13
    a.upload_H(H0)
    a.set_H_scale(scale)
    assert np.array_equal(a.download_H(), Hs)
    a.close(), b.close()


def test_model_fit_queued_loop_equals_blocking_loop_on_the_device():
    """``KLNMF.fit`` with the host out of the loop (queued objectives, speculative kept blocks) against the blocking loop
    (verbose): same stopping iteration, history and bits of W and H -- with a tolerance stop, i.e. through a rollback."""
    X, W0, H0 = orc.synthetic_problem(96, 2000, 8, seed=12)
    fits = []
    for verbose in (0, 1):
        m = sal.models.KLNMF(8, "custom", min_iterations=30, max_iterations=3000, conv_test_freq=10, tol=1e-5)
        m.fit(sal.AnnData(X.copy()), init_kwargs={"signatures_mat": W0.copy(), "exposures_mat": H0.copy()}, verbose=verbose, verbosity_freq=10**9)
        fits.append(m)
    q, b = fits
    assert 30 < q.n_iterations_ < 3000 and q.n_iterations_ == b.n_iterations_
    # (the queued loop's objectives are evaluated inside the update that follows them: another summation order, same value
    # to rounding; the last one, with no update behind it, is the blocking loop's forward pass bit for bit.  The states
    # are the same bits: the update itself does not change.)
    assert np.allclose(q.history["objective_function"], b.history["objective_function"], rtol=1e-12, atol=0)
    assert np.array_equal(q.asignatures.X, b.asignatures.X) and np.array_equal(q.adata.obsm["exposures"], b.adata.obsm["exposures"])
    W, H, it, hist = orc.fit_klnmf(X.T, W0.T, H0.T, min_iterations=30, max_iterations=3000, conv_test_freq=10, tol=1e-5)
    assert it == q.n_iterations_ and np.allclose(q.history["objective_function"], hist, rtol=1e-11)
    # objective_in_step=False: every objective as a pass of its own -- the blocking loop's history bit for bit
    p = sal.models.KLNMF(8, "custom", min_iterations=30, max_iterations=3000, conv_test_freq=10, tol=1e-5, objective_in_step=False)
    p.fit(sal.AnnData(X.copy()), init_kwargs={"signatures_mat": W0.copy(), "exposures_mat": H0.copy()})
    assert p.n_iterations_ == b.n_iterations_ and p.history["objective_function"] == b.history["objective_function"]
    assert np.array_equal(p.asignatures.X, b.asignatures.X) and np.array_equal(p.adata.obsm["exposures"], b.adata.obsm["exposures"])
    assert rel_l2(q.asignatures.X, W.T) < 1e-8 and rel_l2(q.adata.obsm["exposures"], H.T) < 1e-8


# ------------------------------------------------------------------ determinism and split-step (multi-GPU semantics on one GPU)
def test_bitwise_reproducible():
    X, W0, H0 = orc.synthetic_problem(96, 20000, 50, seed=4)
    outs = []
    for _ in range(2):
        e = Engine(20000, 96, 50)
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
        e.kl_step(7)
        outs.append((e.download_W(), e.download_H(), e.objective()))
        e.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]


def test_two_shards_on_one_gpu_equal_unsharded():
    """Two engines hold the two halves of the sample axis; their numerators are summed by hand
    (the all-reduce), both run the W tail: W identical on both and equal to the unsharded run."""
    import torch

    from salamander_amd.distributed import numerator_tensor, shard_bounds

    V, N, K = 96, 5000, 50
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=6)
    rng = np.random.default_rng(2)
    wkl = rng.uniform(0.5, 2, N)
    shards = []
    for r in range(2):
        a, b = shard_bounds(N, 2, r)
        e = Engine(b - a, V, K)
        e.upload_X(X[a:b]), e.upload_W(W0), e.upload_H(H0[a:b])
        e.set_weights(wkl[a:b], None)
        shards.append(e)
    for _ in range(3):
        for e in shards:
            e.kl_step_partial()
            e.sync()
        g0, g1 = numerator_tensor(shards[0]), numerator_tensor(shards[1])
        total = g0 + g1
        g0.copy_(total), g1.copy_(total)
        torch.cuda.synchronize()
        for e in shards:
            e.kl_step_finish(3, _lib.CLIP_ALL)
    W, H = W0.T, H0.T
    for _ in range(3):
        W, H = orc.update_WH(X.T, W, H, wkl, None, 3)
    Ws = [e.download_W() for e in shards]
    assert np.array_equal(Ws[0], Ws[1])
    assert rel_l2(Ws[0], W.T) < TOL
    assert rel_l2(np.concatenate([e.download_H() for e in shards]), H.T) < TOL
    for e in shards:
        e.close()


def test_rccl_communicator_single_rank_path():
    """World size 1 through the in-engine RCCL path (kernel -> ncclAllReduce -> tail) equals the plain path."""
    import os

    import torch.distributed as dist

    from salamander_amd.distributed import attach_communicator

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        X, W0, H0 = orc.synthetic_problem(96, 3000, 50, seed=8)
        res = []
        for with_comm in (False, True):
            e = Engine(3000, 96, 50)
            if with_comm:
                attach_communicator(e)
            e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
            e.kl_step(5, 2)
            res.append((e.download_W(), e.download_H(), e.objective()))
            # the MvNMF step through the same communicator: all-reduce of [G | row sums | KL] and of the trial's
            # KL, next to the two-stream / speculative machinery (identical bits expected at world size 1)
            gamma = e.mv_step(4, 1, 1.0, 1.0, 1.0)
            gamma = e.mv_update_W(0, 1.0, 1.0, gamma)
            res[-1] += (gamma, e.mv_objective(1.0, 1.0), e.download_W(), e.download_H())
            e.close()
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        assert res[0][2] == res[1][2]
        assert res[0][3] == res[1][3] and res[0][4] == res[1][4]
        assert np.array_equal(res[0][5], res[1][5]) and np.array_equal(res[0][6], res[1][6])
    finally:
        if created:
            dist.destroy_process_group()


def test_model_distributed_flag_world_size_one(golden):
    """KLNMF(..., distributed=True).fit on a one-rank process group: communicator attach, W broadcast and
    the kernel -> all-reduce -> tail step give the same fit as the single-GPU path."""
    import os

    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29534")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        g = golden.synth
        fits = []
        for distributed in (False, True):
            adata = sal.AnnData(g["X"].T.copy())
            m = sal.models.KLNMF(50, "custom", min_iterations=30, max_iterations=30, distributed=distributed)
            m.fit(adata, init_kwargs={"signatures_mat": g["W0"].T.copy(), "exposures_mat": g["H0"].T.copy()},
                  fitting_kwargs={"weights_kl": g["wkl"].copy()})
            fits.append((m.asignatures.X.copy(), adata.obsm["exposures"].copy(), list(m.history["objective_function"])))
        assert np.array_equal(fits[0][0], fits[1][0]) and np.array_equal(fits[0][1], fits[1][1])
        assert fits[0][2] == fits[1][2]
        W, H, _, hist = orc.fit_klnmf(g["X"], *_post_init(g), weights_kl=g["wkl"], min_iterations=30, max_iterations=30)
        assert rel_l2(fits[1][0], W.T) < TOL and rel_l2(fits[1][1], H.T) < TOL
        assert np.allclose(fits[1][2], hist, rtol=1e-11)
    finally:
        if created:
            dist.destroy_process_group()


def _post_init(g):
    """The custom init after fit's post-processing (normalise, clip), in reference shapes (V,K), (K,N)."""
    from salamander_amd.initialization import initialize_mat

    S0, E0 = initialize_mat(g["X"].T, g["W0"].shape[1], "custom", signatures_mat=g["W0"].T.copy(), exposures_mat=g["H0"].T.copy())
    return S0.T, E0.T


def test_engine_and_torch_share_one_hip_runtime_in_either_import_order():
    """PyTorch-ROCm bundles its own HIP runtime; of two runtimes in one process the second sees no device.
    ``_lib.load()`` opens torch's HIP runtime first so that both use one copy -- in a fresh process, engine first then torch."""
    import subprocess
    import sys

    code = (
        "from salamander_amd import Engine\n"
        "e = Engine(64, 96, 3)\n"
        "import torch\n"
        "assert torch.cuda.is_available()\n"
        "assert torch.ones(3, device='cuda').sum().item() == 3.0\n"
        "e.close()\n"
        "maps = open('/proc/self/maps').read()\n"
        "assert len({l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}) == 1\n"
        "print('ok')\n"
    )
    from conftest import ROOT

    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout[-500:], r.stderr[-2000:])


def test_engine_first_then_torch_one_rccl_and_a_working_communicator():
    """The multi-rank path's library state, in a fresh process with the engine loaded BEFORE torch: after
    attach_communicator (world size 1, real RCCL communicator inside the engine) exactly one HIP runtime and one RCCL
    are mapped, a sharded step runs, and the process exits without the destructor abort of round 1."""
    import subprocess
    import sys

    code = (
        "import os, numpy as np\n"
        "from salamander_amd import Engine, _lib\n"
        "from salamander_amd.synthetic import synthetic_problem\n"
        "X, W0, H0 = synthetic_problem(96, 2000, 50, seed=1)\n"
        "e = Engine(2000, 96, 50)\n"
        "e.upload_X(X), e.upload_W(W0), e.upload_H(H0)\n"
        "import torch, torch.distributed as dist\n"
        "os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29547')\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "t = torch.ones(4, device='cuda'); dist.all_reduce(t)  # torch's own NCCL backend is live as well\n"
        "from salamander_amd.distributed import attach_communicator\n"
        "attach_communicator(e)\n"
        "e.kl_step(3); W = e.download_W()\n"
        "assert np.all(np.isfinite(W)) and e.comm_info() == (1, 0, 2000)\n"
        "libs = _lib.mapped_runtime_libraries()\n"
        "assert len(libs['libamdhip64']) == 1 and len(libs['librccl']) == 1, libs\n"
        "e.close(); dist.destroy_process_group()\n"
        "print('ok')\n"
    )
    from conftest import ROOT

    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.returncode, r.stdout[-500:], r.stderr[-2000:])


# (K = 1 is left out: with one signature every normalised trial IS the current W, so the line search compares two values
# that differ by rounding only -- the reference's own gamma there is noise, 0.026 after three steps)
@pytest.mark.parametrize("K", [2, 3, 5, 8, 16, 17, 22, 23, 31, 33, 42, 43, 45, 50, 57, 64])
def test_mvnmf_w_only_algebra_every_size(K):
    """The K x K algebra of the MvNMF W step (Gram matrix and the A / B products on the matrix cores, Gauss-Jordan
    elimination in runs of columns per thread: ``csrc/salnmf_mv_device.h``) at every size class of its work split, in the
    spare workgroup of the update_H pass (256 threads: steps inside one call) and in the stand-alone kernels (1024
    threads: single steps, objective): three steps against the oracle, log det through the objective."""
    V, N = 96, 900
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=100 + K)
    lam, delta = 0.7, 0.3
    a, b = Engine(N, V, K), Engine(N, V, K)
    for e in (a, b):
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    assert np.isclose(a.mv_objective(lam, delta), orc.kl_divergence_penalized(X.T, W0.T, H0.T, lam, delta), rtol=1e-11)
    W, H, g = W0.T, H0.T, 1.0
    for _ in range(3):
        W, H, g = orc.mvnmf_step(X.T, W, H, lam, delta, g, 0)
    ga = a.mv_step(3, 0, lam, delta, 1.0)
    gb = 1.0
    for _ in range(3):
        gb = b.mv_step(1, 0, lam, delta, gb)
    assert np.isclose(ga, g, rtol=1e-12) and ga == gb
    assert rel_l2(a.download_W(), W.T) < 1e-9 and rel_l2(a.download_H(), H.T) < 1e-9
    assert np.array_equal(a.download_W(), b.download_W()) and np.array_equal(a.download_H(), b.download_H())
    assert np.isclose(a.mv_objective(lam, delta), orc.kl_divergence_penalized(X.T, W, H, lam, delta), rtol=1e-10)
    a.close(), b.close()


@pytest.mark.parametrize("tag", ["a", "g", "bt1", "bt2"])
def test_mvnmf_function_level_api(golden, tag):
    """The module-level functions of the reference's ``mvnmf.py`` (``volume_logdet``, ``kl_divergence_penalized``,
    ``update_W_unconstrained``, ``line_search``: :19-92) with the reference's argument order and shapes, and the model's
    private hooks ``_update_W_unconstrained`` / ``_line_search`` (:167-188), against the oracle -- on the golden cases,
    two of which backtrack."""
    from salamander_amd.models import mvnmf

    g = golden.mv
    lam, delta, _, ng = g[f"{tag}_par"]
    ng = int(ng)
    X, W, H = g[f"{tag}_X"], g[f"{tag}_W0"], g[f"{tag}_H0"]
    assert np.isclose(mvnmf.volume_logdet(W, delta), orc.volume_logdet(W, delta), rtol=1e-12)
    assert np.isclose(mvnmf.kl_divergence_penalized(X, W, H, lam, delta), orc.kl_divergence_penalized(X, W, H, lam, delta), rtol=1e-11)
    H1 = orc.update_H(X, W, H)
    Wu = mvnmf.update_W_unconstrained(X, W, H1, lam, delta, ng)
    Wu_ref = orc.update_W_unconstrained(X, W, H1, lam, delta, ng)
    # (the K x K inverse by elimination instead of LU: 1e-11 .. 1e-10 on these cases)
    assert Wu.shape == W.shape and rel_l2(Wu, Wu_ref) < 1e-9 and np.array_equal(Wu[:, :ng], W[:, :ng])
    for gamma in (1.0, 0.37):
        Wn, Hn, gn = mvnmf.line_search(X, W, H1, lam, delta, gamma, Wu_ref)
        Wr, Hr, gr = orc.line_search(X, W, H1, lam, delta, gamma, Wu_ref)
        assert np.isclose(gn, gr, rtol=1e-12) and rel_l2(Wn, Wr) < 1e-12 and rel_l2(Hn, Hr) < 1e-12
    # the model's hooks: _update_W == _update_W_unconstrained + _line_search
    a, b = (sal.models.MvNMF(W.shape[1], lam=lam, delta=delta) for _ in range(2))
    for m in (a, b):
        m.adata = sal.AnnData(X.T.copy())
        m.asignatures = sal.AnnData(W.T.copy())
        m.adata.obsm["exposures"] = H1.T.copy()
    a._update_W(ng)
    b._line_search(b._update_W_unconstrained(ng))
    assert a._gamma == b._gamma and rel_l2(b.asignatures.X, a.asignatures.X) < 1e-13
    assert rel_l2(b.adata.obsm["exposures"], a.adata.obsm["exposures"]) < 1e-13


def test_mvnmf_model_fit_with_a_tolerance_stop_matches_the_oracle_fit():
    """``MvNMF.fit`` over many blocks of ``conv_test_freq`` steps: the objectives are the line search's accepted values
    (``mv_step_objective``) and the engine stays ahead between the blocks (``more_follows``) -- same stopping iteration,
    history and factors as the restated reference loop (mvnmf.py:197-210 inside signature_nmf.py:358-385), then a second
    fit on the same model object (no state of the first one may leak)."""
    V, N, K = 96, 3000, 6
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=21)
    kw = dict(min_iterations=20, max_iterations=600, conv_test_freq=10, tol=1e-4)
    lam, delta = 1.0, 1.0
    W, H, _, it, hist = orc.fit_mvnmf(X.T, W0.T, H0.T, lam, delta, **kw)
    assert 20 < it < 600
    m = sal.models.MvNMF(K, "custom", lam, delta, **kw)
    for _ in range(2):
        m.fit(sal.AnnData(X.copy()), init_kwargs={"signatures_mat": W0.copy(), "exposures_mat": H0.copy()})
        # (tolerances of the MvNMF goldens: the K x K inverse by elimination instead of LU moves the trajectory at 1e-9)
        assert m.n_iterations_ == it and np.allclose(m.history["objective_function"], hist, rtol=1e-7)
        assert rel_l2(m.asignatures.X, W.T) < 1e-6 and rel_l2(m.adata.obsm["exposures"], H.T) < 1e-6
        assert np.isclose(m.objective_function(), m.history["objective_function"][-1], rtol=1e-12)  # (a pass of its own vs the step's value)


def test_mvnmf_mixed_call_sequences_match_oracle():
    """State machine check: MvNMF steps leave H lazily rescaled, W / H buffers swapped and (inside a call) the next
    update_H pass possibly pre-computed; every other entry point in between must see the same state as the oracle."""
    V, N, K = 96, 700, 9
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=77)
    lam, delta = 2.0, 0.5

    def mv(W, H, gamma, steps, ng=0):
        for _ in range(steps):
            W, H, gamma = orc.mvnmf_step(X.T, W, H, lam, delta, gamma, ng)
        return W, H, gamma

    e = Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    W, H, g = W0.T, H0.T, 1.0
    # 1. MvNMF steps, then plain KL steps, then MvNMF again
    g_dev = e.mv_step(3, 0, lam, delta, 1.0)
    W, H, g = mv(W, H, g, 3)
    e.kl_step(2, 1)
    for _ in range(2):
        W, H = orc.update_WH(X.T, W, H, None, None, 1)
    g_dev = e.mv_step(2, 2, lam, delta, g_dev)
    W, H, g = mv(W, H, g, 2, ng=2)
    assert np.isclose(g_dev, g, rtol=1e-12)
    assert rel_l2(e.download_W(), W.T) < 1e-8 and rel_l2(e.download_H(), H.T) < 1e-8
    # 2. read-only consumers on a pending rescale (no download in between)
    g_dev = e.mv_step(2, 0, lam, delta, g_dev)
    W, H, g = mv(W, H, g, 2)
    assert np.allclose(e.samplewise_kl(), orc.samplewise_kl_divergence(X.T, W, H), rtol=1e-7)
    assert rel_l2(e.reconstruct(), (W @ H).T) < 1e-8
    assert np.isclose(e.objective(), orc.kl_divergence(X.T, W, H), rtol=1e-9)
    # 3. a stand-alone update_H, then a stand-alone MvNMF W update
    e.update_H()
    H = orc.update_H(X.T, W, H)
    g_dev = e.mv_update_W(0, lam, delta, g_dev)
    Wu = orc.update_W_unconstrained(X.T, W, H, lam, delta, 0)
    W, H, g = orc.line_search(X.T, W, H, lam, delta, g, Wu)
    assert np.isclose(g_dev, g, rtol=1e-12)
    assert rel_l2(e.download_W(), W.T) < 1e-8 and rel_l2(e.download_H(), H.T) < 1e-8
    # 4. new exposures uploaded over a pending rescale, new signatures too
    g_dev = e.mv_step(1, 0, lam, delta, g_dev)
    W, H, g = mv(W, H, g, 1)
    e.upload_H(H0)
    e.upload_W(W0)
    g_dev = e.mv_step(2, 0, lam, delta, g_dev)
    W, H, g = mv(W0.T, H0.T, g, 2)
    assert np.isclose(g_dev, g, rtol=1e-12)
    assert rel_l2(e.download_W(), W.T) < 1e-8 and rel_l2(e.download_H(), H.T) < 1e-8
    e.close()


# ------------------------------------------------------------------ objective accuracy: high counts, perfect fits
def _lane_constants(X):
    """|x log x - x| summed over the six features one lane of the accumulator layout holds: the magnitudes whose
    cancellation against x log p stays inside a lane (salnmf_kernels.h: tile_kl)."""
    Xp = np.zeros((X.shape[0], 96))
    Xp[:, : X.shape[1]] = X
    t = np.where(Xp > 0, Xp * np.log(np.where(Xp > 0, Xp, 1.0)) - Xp, 0.0)
    return np.abs(t.reshape(X.shape[0], 6, 16).sum(axis=1))


@pytest.mark.parametrize("N,V,K,mean", [(16 * 1074 - 5, 96, 50, 5e7), (5003, 83, 12, 2e8), (20000, 96, 30, 3e6)])
def test_objective_of_high_count_catalogues_matches_the_per_entry_form(N, V, K, mean):
    """Counts of 1e5-1e6 per entry: x log x and x log p are ~1e7 each and cancel to a KL term of order 1.  The engine splits
    the two (one logarithm per entry) but cancels them inside a lane, per (sample, lane column); the oracle evaluates the
    reference's per-entry form x log(x / p) - x + p (``_utils_klnmf.py:41-53``).  rtol 1e-10 on every objective path: the
    forward pass, the objective folded into a step, the weighted objective, MvNMF's penalised objective (sizes with a
    cooperative leftover tile, ragged N, V < 96)."""
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=3, mean_mutations=mean)
    assert X.max() > 1e5
    e = Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.kl_step(3, 0)
    W, H = e.download_W(), e.download_H()
    want = orc.kl_divergence(X.T, W.T, H.T)
    assert np.isclose(e.objective(), want, rtol=1e-10, atol=0)
    e.kl_step_objective(0, 1, 0, keep=True)  # the same state's objective inside a step
    e.kl_rollback()
    assert np.isclose(e.objective_read(0, 1)[0], want, rtol=1e-10, atol=0)
    assert np.isclose(e.mv_objective(1.0, 1.0), orc.kl_divergence_penalized(X.T, W.T, H.T, 1.0, 1.0), rtol=1e-10, atol=0)
    wk = np.random.default_rng(1).uniform(0.5, 2.0, N)
    e.set_weights(wk, None)
    assert np.isclose(e.objective(), orc.kl_divergence(X.T, W.T, H.T, wk), rtol=1e-10, atol=0)
    got = e.samplewise_kl()
    assert np.allclose(got, orc.samplewise_kl_divergence(X.T, W.T, H.T), rtol=1e-9, atol=0) and got.min() >= 0.0
    e.close()


@pytest.mark.parametrize("N,V,K,mean", [(4099, 96, 50, 2000.0), (3000, 96, 8, 1e7)])
def test_objective_of_a_perfect_fit_is_zero_to_the_rounding_of_a_lane(N, V, K, mean):
    """X = W H to rounding: the KL divergence is 0 (the reference's per-entry form gives ~1e-16 sum |x|).  The split form
    leaves the rounding of x log x against x log p: at most a few eps times the magnitude held by one lane, summed in
    quadrature over the lanes -- not eps times the sum over the whole matrix, which a constant per sample or per matrix
    would leave (4-7 digits more at these sizes).  The per-sample divergences keep the per-entry form: >= -1e-12 sum_v x."""
    _, W0, H0 = orc.synthetic_problem(V, N, K, seed=5, mean_mutations=mean)
    X = H0 @ W0  # (N, V): not counts any more, but exactly representable products are not needed: P differs by rounding
    e = Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    c = _lane_constants(X)
    bound = 32 * np.finfo(float).eps * np.sqrt((c**2).sum())
    for value in (e.objective(), e.mv_objective(0.0, 1.0)):
        assert abs(value) <= bound, (value, bound, np.finfo(float).eps * c.sum())
    e.kl_step_objective(0, 1, 0, keep=True)
    e.kl_rollback()
    assert abs(e.objective_read(0, 1)[0]) <= bound
    errs = e.samplewise_kl()
    assert errs.min() >= -1e-12 * X.sum(axis=1).max() and errs.max() <= 1e-9 * X.sum(axis=1).max()
    e.close()


# ------------------------------------------------------------------ MvNMF: steps queued ahead of the host vs the classic form
@pytest.mark.parametrize("tag", ["a", "g", "bt1", "bt2"])
def test_mvnmf_queued_steps_equal_the_classic_form_bit_for_bit(golden, tag):
    """Default: per step a tail and ONE pass over the samples (``fused_kernel<.., MVJ>``), the line-search decision on the
    device, the host reading a flag once per call.  ``set_mv_queued(False)``: the classic form (two passes per step, the
    host decides every step).  Same gammas, W, H and objectives bit for bit -- in one call, step by step, in blocks that
    leave the engine ahead between calls; ``bt1`` / ``bt2`` reject first trials in the middle of a queued batch."""
    g = golden.mv
    lam, delta, steps, ng = g[f"{tag}_par"]
    steps, ng = int(steps), int(ng)
    outs = []
    for queued in (True, False):
        res = []
        # (1) all steps in one call; (2) blocks of 2 with more_follows; (3) step by step
        for mode in ("one", "blocks", "single"):
            e = make_engine(g[f"{tag}_X"], g[f"{tag}_W0"], g[f"{tag}_H0"])
            e.set_mv_queued(queued)
            gamma, objs = 1.0, []
            if mode == "one":
                gamma = e.mv_step(steps, ng, lam, delta, gamma)
            elif mode == "blocks":
                left = steps
                while left > 0:
                    n = min(2, left)
                    gamma, f = e.mv_step_objective(n, ng, lam, delta, gamma, more_follows=left > n)
                    objs.append(f)
                    left -= n
            else:
                for _ in range(steps):
                    gamma = e.mv_step(1, ng, lam, delta, gamma)
            res.append((gamma, e.download_W(), e.download_H(), objs, e.mv_objective(lam, delta)))
            e.close()
        outs.append(res)
    for (gq, Wq, Hq, oq, fq), (gc, Wc, Hc, oc, fc) in zip(*outs):
        assert gq == gc and np.array_equal(Wq, Wc) and np.array_equal(Hq, Hc) and oq == oc and fq == fc
    # the three ways of splitting the steps agree among themselves, and with the golden trajectory
    ref = outs[0][0]
    for r in outs[0][1:]:
        assert r[0] == ref[0] and np.array_equal(r[1], ref[1]) and np.array_equal(r[2], ref[2])
    assert np.isclose(ref[0], g[f"{tag}_gammas"][-1], rtol=1e-12) and rel_l2(ref[1], g[f"{tag}_W"].T) < 1e-7


@pytest.mark.parametrize("tag", ["bt1", "bt2"])
def test_mvnmf_queued_form_edge_cases_equal_the_classic_form(golden, tag):
    """the queued form's host bookkeeping at its corners, against the classic form bit for bit -- (i) a batch
    whose FIRST queued step is rejected with more steps queued behind it (one step, then the rest in one call); (ii) a
    rejection on the LAST queued step of a call that leaves the engine ahead (``more_follows``), found by cutting the
    trajectory right behind every step whose gamma dropped; (iii) an entry gamma of 1e-17, where ``gamma > 1e-16``
    (mvnmf.py:84) overrides the device's verdict and every trial is accepted."""
    g = golden.mv
    lam, delta, steps, ng = g[f"{tag}_par"]
    steps, ng = int(steps), int(ng)
    gam = g[f"{tag}_gammas"]
    drops = [i for i in range(steps) if gam[i] < min(1.0, 1.2 * (gam[i - 1] if i else 1.0))]  # steps that backtracked
    assert drops
    plans = [[1, steps - 1]]                                    # (i)
    plans += [[d + 1, steps - d - 1] for d in drops if 0 < d + 1 < steps]  # (ii): the call ends ON a rejected step, engine left ahead
    for plan in plans:
        outs = []
        for queued in (True, False):
            e = make_engine(g[f"{tag}_X"], g[f"{tag}_W0"], g[f"{tag}_H0"])
            e.set_mv_queued(queued)
            gamma, fs, left = 1.0, [], steps
            for n in plan:
                left -= n
                gamma, f = e.mv_step_objective(n, ng, lam, delta, gamma, more_follows=left > 0)
                fs.append(f)
            outs.append((gamma, fs, e.download_W(), e.download_H()))
            e.close()
        (gq, fq, Wq, Hq), (gc, fc, Wc, Hc) = outs
        assert gq == gc and fq == fc and np.array_equal(Wq, Wc) and np.array_equal(Hq, Hc), plan
        assert np.isclose(gq, gam[-1], rtol=1e-12) and rel_l2(Wq, g[f"{tag}_W"].T) < 1e-7
    # (iii) gamma <= 1e-16 on entry: no trial can be rejected
    outs = []
    for queued in (True, False):
        e = make_engine(g[f"{tag}_X"], g[f"{tag}_W0"], g[f"{tag}_H0"])
        e.set_mv_queued(queued)
        g1, f1 = e.mv_step_objective(3, ng, lam, delta, 1e-17, more_follows=True)
        g2, f2 = e.mv_step_objective(3, ng, lam, delta, g1)
        outs.append((g1, g2, f1, f2, e.download_W(), e.download_H()))
        e.close()
    assert outs[0][:4] == outs[1][:4] and np.array_equal(outs[0][4], outs[1][4]) and np.array_equal(outs[0][5], outs[1][5])
    W, H, gg = g[f"{tag}_W0"], g[f"{tag}_H0"], 1e-17
    for _ in range(6):
        W, H, gg = orc.mvnmf_step(g[f"{tag}_X"], W, H, lam, delta, gg, ng)
    assert np.isclose(outs[0][1], gg, rtol=1e-12) and rel_l2(outs[0][4], W.T) < 1e-7


@pytest.mark.parametrize("N,K", [(17003, 33), (900, 17), (40000, 50), (5000, 8)])
def test_mvnmf_queued_steps_on_more_shapes(N, K):
    X, W0, H0 = orc.synthetic_problem(96, N, K, seed=N % 97)
    lam, delta = 1.0, 1.0
    res = []
    for queued in (True, False):
        e = Engine(N, 96, K)
        e.set_mv_queued(queued)
        e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
        gamma, f = e.mv_step_objective(6, 0, lam, delta, 1.0, more_follows=True)
        gamma, f2 = e.mv_step_objective(5, 0, lam, delta, gamma, more_follows=False)
        res.append((gamma, f, f2, e.download_W(), e.download_H()))
        e.close()
    assert res[0][:3] == res[1][:3] and np.array_equal(res[0][3], res[1][3]) and np.array_equal(res[0][4], res[1][4])
    W, H, g = W0.T, H0.T, 1.0
    for _ in range(11):
        W, H, g = orc.mvnmf_step(X.T, W, H, lam, delta, g, 0)
    assert np.isclose(res[0][0], g, rtol=1e-12) and rel_l2(res[0][3], W.T) < 1e-7 and rel_l2(res[0][4], H.T) < 1e-7
