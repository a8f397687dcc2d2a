"""Host-side logic on CPU: checkers, AnnData shim, initialisation, fit-loop cadence, hooks.

The device is replaced by the oracle-backed ``FakeEngine`` (tests only), so what is tested
here is the Python mirror of the reference's operator interface, not the kernels.
"""

import os

import numpy as np
import pandas as pd
import pytest

import salamander_amd as sal
from _fake_engine import FakeEngine
from conftest import REF_FIX, read_counts, rel_l2
from oracle import klnmf_oracle as orc
from salamander_amd import initialization as init
from salamander_amd import utils
from salamander_amd.anndata_compat import MiniAnnData
from salamander_amd.models import signature_nmf


@pytest.fixture
def fake_engine(monkeypatch):
    monkeypatch.setattr(signature_nmf, "Engine", FakeEngine)
    return FakeEngine


@pytest.fixture
def counts():
    return read_counts(os.path.join(REF_FIX, "klnmf", "counts.csv"))


def make_adata(counts):
    return sal.AnnData(counts.T)


# ---------------------------------------------------------------- checkers (utils.py:16-99)
def test_checkers_raise_like_the_reference():
    with pytest.raises(TypeError):
        utils.type_checker("x", 1.0, int)
    utils.type_checker("x", 1, [int, float])
    with pytest.raises(TypeError):
        utils.type_checker("x", np.float64(1.0), [float])  # exact type, not isinstance
    with pytest.raises(ValueError):
        utils.shape_checker("a", np.zeros((2, 3)), (3, 2))
    with pytest.raises(TypeError):
        utils.shape_checker("a", [[1]], (1, 1))
    with pytest.raises(ValueError):
        utils.value_checker("m", "foo", ["a", "b"])
    with pytest.raises(ValueError):
        utils.dict_checker("d", {"z": 1}, ["a"])
    with pytest.raises(TypeError):
        utils.dict_checker("d", [("a", 1)], ["a"])


def test_normalize_WH():
    rng = np.random.default_rng(0)
    W, H = rng.random((96, 4)), rng.random((4, 7))
    Wn, Hn = utils.normalize_WH(W, H)
    assert np.allclose(Wn.sum(0), 1) and np.allclose(Wn @ Hn, W @ H)
    Wo, Ho = orc.normalize_WH(W, H)
    assert np.array_equal(Wn, Wo) and np.array_equal(Hn, Ho)


# ---------------------------------------------------------------- AnnData shim
def test_mini_anndata_surface(counts):
    a = MiniAnnData(counts.T)
    assert (a.n_obs, a.n_vars) == (10, 96)
    assert list(a.obs_names) == list(counts.columns) and list(a.var_names) == list(counts.index)
    a.obsm["exposures"] = np.ones((10, 2))
    a.obs["reconstruction_error"] = np.arange(10.0)
    b = a.copy()
    b.X[0, 0] = -1
    assert a.X[0, 0] != -1
    sub = a[:3, :]
    assert sub.n_obs == 3 and sub.obsm["exposures"].shape == (3, 2) and list(sub.obs_names) == list(counts.columns[:3])
    assert a.to_df().shape == (10, 96)
    with pytest.raises(ValueError):
        a.X = np.zeros((3, 3))


# ---------------------------------------------------------------- initialisation (methods.py, initialize.py)
def test_init_methods_shapes_and_postprocessing(counts):
    X = counts.T.values.astype(float)
    for method, kw in (("flat", {}), ("random", {"seed": 1}), ("separableNMF", {"seed": 1}), ("nndsvda", {"seed": 1})):
        S, E = init.initialize_mat(X, 3, method, **kw)
        assert S.shape == (3, 96) and E.shape == (10, 3)
        assert (S >= utils.EPSILON).all() and (E >= utils.EPSILON).all()
        assert np.allclose(S.sum(1), 1, atol=1e-4)
    S1, E1 = init.initialize_mat(X, 3, "random", seed=5)
    S2, E2 = init.initialize_mat(X, 3, "random", seed=5)
    assert np.array_equal(S1, S2) and np.array_equal(E1, E2)


def test_init_random_matches_oracle_restatement(counts):
    """init_random + normalise + clip equals what the reference computes from the same global RNG stream."""
    X = counts.T.values.astype(float)
    S, E = init.initialize_mat(X, 2, "random", seed=3)
    np.random.seed(3)
    S0 = np.random.dirichlet(np.ones(96), size=2)
    E0 = X.sum(1)[:, None] * np.random.dirichlet(np.ones(2), size=10)
    W, H = orc.normalize_WH(S0.T, E0.T)
    assert np.allclose(S, W.T.clip(orc.EPSILON)) and np.allclose(E, H.T.clip(orc.EPSILON))


@pytest.mark.parametrize("method", ["flat", "nndsvd", "nndsvda", "nndsvdar", "random", "separableNMF"])
def test_init_methods_match_reference_fixtures(method):
    """The reference's own initialisation fixtures (tests/test_initialization.py:28-53), seed 1, K = 2."""
    d = os.path.join(REF_FIX, "initialization")
    data = np.load(f"{d}/data_mat.npy")
    suffix = "flat.npy" if method == "flat" else f"{method}_seed1.npy"
    kwargs = {} if method == "flat" else {"seed": 1}
    S, E = init.initialize_mat(data, 2, method, **kwargs)
    assert np.allclose(S, np.load(f"{d}/signatures_mat_{suffix}"))
    assert np.allclose(E, np.load(f"{d}/exposures_mat_{suffix}"))


@pytest.mark.parametrize("method", ["flat", "nndsvd", "nndsvda"])
def test_device_init_driver_matches_reference_fixtures(method):
    """``device_init.initialize_on_device`` (exact SVD through the Gram matrix; here on the NumPy stand-in of the
    engine primitives) against the reference's fixtures: at 96 x 10, K = 2 the randomized SVD behind them is exact,
    and NNDSVD does not depend on the signs of the singular vectors, so the entries agree."""
    from salamander_amd.device_init import initialize_on_device

    d = os.path.join(REF_FIX, "initialization")
    data = np.load(f"{d}/data_mat.npy")
    suffix = "flat.npy" if method == "flat" else f"{method}_seed1.npy"
    e = FakeEngine(data.shape[0], data.shape[1], 2)
    e.upload_X(data)
    S = initialize_on_device(e, 2, method)
    assert np.allclose(S, np.load(f"{d}/signatures_mat_{suffix}"), rtol=1e-7, atol=1e-12)
    assert np.allclose(e.download_H(), np.load(f"{d}/exposures_mat_{suffix}"), rtol=1e-7, atol=1e-12)


def test_model_default_init_runs_on_the_device_unless_seeded(fake_engine, counts):
    """KLNMF's default nndsvd goes through the engine's init primitives; a seed (or device_init=False) selects the
    reference's host computation; both leave the same kind of state behind."""
    host = sal.models.KLNMF(2, "nndsvd", min_iterations=1, max_iterations=1)
    host.fit(make_adata(counts), init_kwargs={"seed": 1})
    dev = sal.models.KLNMF(2, "nndsvd", min_iterations=1, max_iterations=1)
    dev.fit(make_adata(counts))
    off = sal.models.KLNMF(2, "nndsvd", min_iterations=1, max_iterations=1, device_init=False)
    np.random.seed(1)
    off.fit(make_adata(counts))
    assert rel_l2(dev.asignatures.X, host.asignatures.X) < 1e-6 and rel_l2(off.asignatures.X, host.asignatures.X) < 1e-12
    assert list(dev.asignatures.obs_names) == ["Sig1", "Sig2"] and dev.adata.obsm["exposures"].shape == (10, 2)
    with pytest.raises(TypeError):
        sal.models.KLNMF(2, "flat").fit(make_adata(counts), init_kwargs={"bogus": 1})


def test_init_custom_keeps_normalised_input_bitwise():
    """tests/test_initialization.py:56-67: a normalised custom init comes back untouched."""
    data = np.load(os.path.join(REF_FIX, "initialization", "data_mat.npy"))
    S0 = np.array([[0.1, 0.2, 0.7], [0.6, 0.1, 0.3]])
    E0 = np.arange(1, 9).reshape((4, 2))
    S, E = init.initialize_mat(data, 2, "custom", signatures_mat=S0, exposures_mat=E0)
    assert np.array_equal(S0, S) and np.array_equal(E0, E)


def test_init_custom_and_errors(counts):
    X = counts.T.values.astype(float)
    S0, E0 = np.full((2, 96), 1 / 96), np.ones((10, 2))
    S, E = init.initialize_mat(X, 2, "custom", signatures_mat=S0.copy(), exposures_mat=E0.copy())
    assert np.allclose(S, S0) and np.allclose(E, E0)
    with pytest.raises(ValueError):
        init.initialize_mat(X, 2, "custom", signatures_mat=S0[:, :5], exposures_mat=E0)
    with pytest.raises(TypeError):
        init.initialize_mat(X, 2, "custom", signatures_mat=S0.tolist(), exposures_mat=E0)
    with pytest.raises(ValueError):
        init.initialize_mat(X, 2, "bogus")
    with pytest.raises(ValueError):
        init.initialize_mat(X, 1, "flat", given_signatures_mat=np.ones((2, 96)))


def test_given_signatures_keep_names_and_values(counts):
    adata = make_adata(counts)
    given = adata[:1, :].copy()
    given.X = given.X / given.X.sum(axis=1, keepdims=True)
    asig = init.initialize_standard_nmf(adata, 3, "flat", {"asignatures": given})
    assert list(asig.obs_names) == [given.obs_names[0], "Sig1", "Sig2"]
    assert np.allclose(np.asarray(asig.X)[0], given.X[0])
    assert adata.obsm["exposures"].shape == (10, 3)
    with pytest.raises(ValueError):
        init.initialize_standard_nmf(adata, 3, "flat", {"bogus": 1})
    bad = given.copy()
    bad.var_names = [f"f{i}" for i in range(96)]
    with pytest.raises(ValueError):
        init.initialize_standard_nmf(adata, 3, "flat", {"asignatures": bad})


# ---------------------------------------------------------------- model API surface
def test_constructor_defaults_match_reference():
    m = sal.models.KLNMF()
    assert (m.n_signatures, m.init_method, m.min_iterations, m.max_iterations, m.conv_test_freq, m.tol) == (
        1, "nndsvd", 500, 10000, 10, 1e-7)
    mv = sal.models.MvNMF(3, "random", 2.0, 0.5)
    assert (mv.lam, mv.delta, mv._gamma, mv.objective) == (2.0, 0.5, 1.0, "minimize")
    with pytest.raises(ValueError):
        sal.models.KLNMF(init_method="nope")


def test_fit_rejects_non_anndata(fake_engine):
    with pytest.raises(TypeError):
        sal.models.KLNMF(2).fit(np.ones((10, 96)))


def test_fitting_kwargs_validation(fake_engine, counts):
    adata = make_adata(counts)
    m = sal.models.KLNMF(2, "flat", min_iterations=1, max_iterations=1)
    with pytest.raises(ValueError):
        m.fit(adata.copy(), fitting_kwargs={"weights": 1.0})
    with pytest.raises(ValueError):
        m.fit(adata.copy(), fitting_kwargs={"weights_kl": -1.0})
    with pytest.raises(ValueError):
        m.fit(adata.copy(), fitting_kwargs={"weights_kl": np.ones(3)})
    with pytest.raises(TypeError):
        m.fit(adata.copy(), fitting_kwargs={"weights_kl": "heavy"})
    m.fit(adata.copy(), fitting_kwargs={"weights_kl": 2.0, "weights_lhalf": [0.5] * 10})
    assert np.array_equal(m.weights_kl, 2 * np.ones(10)) and np.array_equal(m.weights_lhalf, 0.5 * np.ones(10))


def test_fit_clips_callers_adata_and_writes_results(fake_engine, counts):
    adata = make_adata(counts)
    adata.X = adata.X.astype(float)
    adata.X[0, 0] = 0.0
    m = sal.models.KLNMF(2, "random", min_iterations=20, max_iterations=20)
    out = m.fit(adata, init_kwargs={"seed": 1})
    assert out is m
    assert adata.X[0, 0] == utils.EPSILON  # the caller's object is mutated (signature_nmf.py:281)
    assert m.asignatures.X.shape == (2, 96) and adata.obsm["exposures"].shape == (10, 2)
    assert list(m.asignatures.obs_names) == ["Sig1", "Sig2"]
    assert m.signatures.shape == (2, 96) and m.exposures.shape == (10, 2)
    assert len(m.history["objective_function"]) == 2
    assert m.data_reconstructed.shape == (10, 96)
    assert m.reconstruction_error > 0 and "reconstruction_error" in adata.obs


@pytest.mark.parametrize(
    "min_it,max_it,freq,expect_chunks",
    [
        (30, 30, 10, [10, 10, 10]),
        (25, 25, 10, [10, 10, 5]),
        (3, 3, 10, [3]),
        (12, 12, 5, [5, 5, 2]),
        # the reference tests `n_iteration >= max_iterations` after the first update: a cap <= 0 still runs one
        (0, 0, 10, [1]),
        (0, -5, 10, [1]),
    ],
)
def test_fit_loop_chunks_and_history_cadence(fake_engine, counts, min_it, max_it, freq, expect_chunks):
    adata = make_adata(counts)
    m = sal.models.KLNMF(2, "flat", min_iterations=min_it, max_iterations=max_it, conv_test_freq=freq)
    m.fit(adata)
    assert m._engine.steps_log == expect_chunks
    assert m.n_iterations_ == max(max_it, 1)
    assert len(m.history["objective_function"]) == max(max_it, 1) // freq


def test_fit_equals_oracle_fit_including_convergence_stop(fake_engine, counts):
    """Same stop iteration, W, H and history as the restated reference loop (tolerance-triggered stop)."""
    X = counts.T.values.astype(float)
    S0, E0 = init.initialize_mat(X.clip(utils.EPSILON), 2, "random", seed=2)
    kw = dict(min_iterations=20, max_iterations=400, conv_test_freq=10, tol=1e-4)
    W, H, it, hist = orc.fit_klnmf(X.T, S0.T, E0.T, **kw)
    assert it < 400  # really stopped by the tolerance
    m = sal.models.KLNMF(2, "custom", **kw)
    m.fit(make_adata(counts), init_kwargs={"signatures_mat": S0.copy(), "exposures_mat": E0.copy()})
    assert m.n_iterations_ == it
    assert np.allclose(m.history["objective_function"], hist, rtol=1e-13)
    assert rel_l2(m.asignatures.X, W.T) < 1e-13 and rel_l2(m.adata.obsm["exposures"], H.T) < 1e-13


def test_fit_queues_objectives_until_min_iterations_then_speculates(fake_engine, counts):
    """The host is out of the loop: before ``min_iterations`` no objective is read back -- they are
    queued into the ring and fetched in one read; from then on the next block is launched (as a kept block) before the
    deciding objective is read, and rolled back when the test says "converged".  Same history, stopping iteration and
    final state as the blocking loop (which verbose fits still use)."""
    X = counts.T.values.astype(float)
    S0, E0 = init.initialize_mat(X.clip(utils.EPSILON), 2, "random", seed=2)
    kw = dict(min_iterations=40, max_iterations=400, conv_test_freq=10, tol=1e-4)
    init_kw = lambda: {"signatures_mat": S0.copy(), "exposures_mat": E0.copy()}
    q = sal.models.KLNMF(2, "custom", **kw)
    q.fit(make_adata(counts), init_kwargs=init_kw())
    e = q._engine
    assert getattr(e, "blocking_objectives", 0) == 0
    # steps up to min_iterations are plain launches, every block after a decision point is a kept one, and the last
    # kept block was discarded
    assert e.steps_log[:4] == [10, 10, 10, 10] and all(s == ("keep", 10) for s in e.steps_log[4:-1]) and e.steps_log[-1] == "rollback"
    # one read fetched everything queued before the first decision (initial objective + 4 checks); then one per decision
    assert e.reads[0] == (0, 5) and all(c == 1 for _, c in e.reads[1:])
    assert q.n_iterations_ < 400 and q.n_iterations_ == 10 * (len(e.steps_log) - 2)
    b = sal.models.KLNMF(2, "custom", **kw)
    b.fit(make_adata(counts), init_kwargs=init_kw(), verbose=1, verbosity_freq=10**9)
    assert b._engine.blocking_objectives == len(b.history["objective_function"]) + 1
    assert q.n_iterations_ == b.n_iterations_ and q.history["objective_function"] == b.history["objective_function"]
    assert np.array_equal(q.asignatures.X, b.asignatures.X) and np.array_equal(q.adata.obsm["exposures"], b.adata.obsm["exposures"])


def test_a_fit_that_fails_midway_leaves_the_callers_exposures_in_place(fake_engine, counts, monkeypatch):
    """The queued loop takes the host copy of the initial exposures out of the caller's AnnData while the device works
    (they come back fitted).  An engine error -- or an interrupt -- after that point must not leave the object without
    them: the reference never removes ``obsm["exposures"]`` (``initialize.py:254``, ``klnmf.py:106``)."""
    adata = make_adata(counts)
    m = sal.models.KLNMF(2, "random", min_iterations=40, max_iterations=400, conv_test_freq=10, tol=1e-12)
    calls = {"n": 0}
    real = fake_engine.objective_read

    def failing_read(self, first, count):
        calls["n"] += 1
        if calls["n"] >= 3:
            raise RuntimeError("device lost")
        return real(self, first, count)

    monkeypatch.setattr(fake_engine, "objective_read", failing_read)
    with pytest.raises(RuntimeError, match="device lost"):
        m.fit(adata, init_kwargs={"seed": 1})
    assert "exposures" in adata.obsm and adata.obsm["exposures"].shape == (10, 2) and np.isfinite(adata.obsm["exposures"]).all()
    assert adata.X.min() >= utils.EPSILON  # the clipped matrix is in place as after any fit (signature_nmf.py:281)


@pytest.mark.parametrize(
    "min_it,max_it,freq,tol",
    [(0, 5, 10, 1e-4), (1, 1, 1, 1e-4), (0, 0, 10, 1e-4), (5, 37, 10, 1e-9), (40, 400, 10, 1e-4), (40, 400, 7, 1e-3), (0, 400, 1, 1e-4),
     (30, 30, 10, 1e-4), (25, 1000, 10, 1e-2), (10, 10000, 10, 1e-3), (3, 50, 3, 0.0)],
)
def test_queued_loop_equals_blocking_loop_over_the_parameter_space(fake_engine, counts, min_it, max_it, freq, tol):
    """signature_nmf.py:358-385 has corner cases (cap below the test frequency, a cap that is no multiple of it, a zero
    cap, tolerance hit at the first permitted test): the queued loop -- objectives folded into the following block,
    speculative kept blocks, rollback -- stops at the same iteration with the same history and factors as the blocking one."""
    X = counts.T.values.astype(float)
    S0, E0 = init.initialize_mat(X.clip(utils.EPSILON), 2, "random", seed=5)
    kw = dict(min_iterations=min_it, max_iterations=max_it, conv_test_freq=freq, tol=tol)
    fits = []
    for verbose in (0, 1):
        m = sal.models.KLNMF(2, "custom", **kw)
        m.fit(make_adata(counts), init_kwargs={"signatures_mat": S0.copy(), "exposures_mat": E0.copy()}, verbose=verbose, verbosity_freq=10**9)
        fits.append(m)
    q, b = fits
    assert q.n_iterations_ == b.n_iterations_ and q.history["objective_function"] == b.history["objective_function"]
    assert np.array_equal(q.asignatures.X, b.asignatures.X) and np.array_equal(q.adata.obsm["exposures"], b.adata.obsm["exposures"])
    assert getattr(q._engine, "blocking_objectives", 0) == 0
    W, H, it, hist = orc.fit_klnmf(X.T, S0.T, E0.T, **kw)
    assert it == q.n_iterations_ and np.allclose(q.history["objective_function"], hist, rtol=1e-13)


def test_fit_objective_ring_wraps(fake_engine, counts):
    """More queued objectives than the device ring has slots (256): the loop drains the ring when it is full."""
    m = sal.models.KLNMF(2, "flat", min_iterations=1100, max_iterations=1100, conv_test_freq=1)
    m.fit(make_adata(counts))
    hist = m.history["objective_function"]
    assert len(hist) == 1100 and np.all(np.isfinite(hist)) and np.all(np.diff(hist) <= 1e-9 * np.abs(hist[:-1]))
    assert m._engine.reads[0] == (0, 256) and sum(c for _, c in m._engine.reads) == 1101


def test_verbose_prints_at_the_reference_iterations(fake_engine, counts, capsys):
    m = sal.models.KLNMF(2, "flat", min_iterations=25, max_iterations=25)
    m.fit(make_adata(counts), verbose=1, verbosity_freq=7)
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("iteration")]
    assert [int(l.split(";")[0].split(":")[1]) for l in lines] == [7, 14, 21]
    assert sum(m._engine.steps_log) == 25


def test_single_step_hooks_on_hand_populated_model(fake_engine, counts):
    """tests/test_klnmf.py:44-75 drives _update_parameters()/objective_function() without fit()."""
    d = os.path.join(REF_FIX, "klnmf")
    W0, H0 = np.load(f"{d}/W_init_nsigs2.npy"), np.load(f"{d}/H_init_nsigs2.npy")
    m = sal.models.KLNMF(n_signatures=2)
    m.adata = make_adata(counts)
    m.asignatures = sal.AnnData(W0.T)
    m.adata.obsm["exposures"] = H0.T
    assert np.allclose(m.objective_function(), np.load(f"{d}/objective_init_nsigs2.npy"))
    m._update_parameters()
    W1, H1 = orc.update_WH(counts.values.astype(float), W0, H0)
    assert np.allclose(m.asignatures.X, W1.T) and np.allclose(m.adata.obsm["exposures"], H1.T)


def test_mvnmf_gamma_persists_and_resets(fake_engine, counts, golden):
    # a configuration whose line search backtracks (tests/golden/make_golden.py, case bt1)
    g = golden.mv
    lam, delta, steps, _ = g["bt1_par"]
    X, W0, H0 = g["bt1_X"], g["bt1_W0"], g["bt1_H0"]
    adata = sal.AnnData(X.T.copy())
    m = sal.models.MvNMF(W0.shape[1], "custom", lam, delta, min_iterations=int(steps), max_iterations=int(steps))
    m._gamma = 0.3  # stale state from an earlier fit must be reset (mvnmf.py:218)
    m.fit(adata, init_kwargs={"signatures_mat": W0.T.copy(), "exposures_mat": H0.T.copy()})
    assert np.isclose(m._gamma, g["bt1_gammas"][-1], rtol=1e-12) and m._gamma < 1.0
    assert rel_l2(m.asignatures.X, g["bt1_W"].T) < 1e-8
    # all signatures given: W is never touched and gamma stays at its reset value
    given = make_adata(counts)[:2, :].copy()
    given.X = given.X / given.X.sum(axis=1, keepdims=True)
    m2 = sal.models.MvNMF(2, "flat", min_iterations=3, max_iterations=3)
    m2.fit(make_adata(counts), given_parameters={"asignatures": given})
    assert np.allclose(m2.asignatures.X, given.X) and m2._gamma == 1.0


def test_out_of_scope_helpers_say_so():
    with pytest.raises(NotImplementedError):
        sal.models.KLNMF().plot_signatures()


def test_package_synthetic_generator_equals_the_oracles():
    """bench.py and tools draw inputs from salamander_amd.synthetic; tests from the oracle: same data for a seed."""
    from oracle import klnmf_oracle as orc
    from salamander_amd.synthetic import synthetic_problem

    for V, N, K, seed in [(96, 257, 7, 3), (83, 100, 50, 0)]:
        for a, b in zip(synthetic_problem(V, N, K, seed=seed), orc.synthetic_problem(V, N, K, seed=seed)):
            assert np.array_equal(a, b)
