"""GPU parity for more than 64 signatures (VERDICT r2, item 7 / "what's missing" 3): n_signatures in {65 .. 512}.

The reference has no limit on the number of signatures (``_utils_klnmf.py:281-361``); the engine runs such problems per
chunk of <= 64 signatures: the product H W is accumulated over the chunks by a chain of forward launches, the last of
which forms the ratio X / (H W) (or the divergence, the per-sample divergences, the reconstruction), and the update passes
run once per chunk on that ratio.  Every KLNMF entry point against the oracle on ragged N, with weights, l-half
penalties, given signatures (inside and across chunks) and zeros in X."""

import numpy as np
import pytest

import salamander_amd as sal
from conftest import rel_l2
from oracle import klnmf_oracle as orc
from salamander_amd import Engine, _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V,N,K", [(96, 1000, 65), (96, 2049, 100), (83, 5003, 128), (96, 777, 150), (20, 333, 200), (96, 1200, 512)])
def test_many_signatures_functions_against_the_oracle(V, N, K):
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=V + N)
    e = Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    assert np.array_equal(e.download_W(), W0) and np.array_equal(e.download_H(), H0)
    assert np.isclose(e.objective(), orc.kl_divergence(X.T, W0.T, H0.T), rtol=1e-12)
    assert np.allclose(e.samplewise_kl(), orc.samplewise_kl_divergence(X.T, W0.T, H0.T), rtol=1e-11)
    assert rel_l2(e.reconstruct(), H0 @ W0) < 1e-14
    e.update_H()
    H1 = orc.update_H(X.T, W0.T, H0.T).T
    assert rel_l2(e.download_H(), H1) < 1e-13
    e.upload_H(H0)
    e.update_W(2, _lib.CLIP_NON_GIVEN)
    W1 = orc.update_W(X.T, W0.T, H0.T, n_given_signatures=2).T
    assert rel_l2(e.download_W(), W1) < 1e-13 and np.array_equal(e.download_W()[:2], W0[:2])
    e.upload_W(W0)
    W, H = W0.T, H0.T
    for _ in range(5):
        W, H = orc.update_WH(X.T, W, H)
    e.kl_step(5)
    assert rel_l2(e.download_W(), W.T) < 1e-12 and rel_l2(e.download_H(), H.T) < 1e-12
    assert np.isclose(e.objective(), orc.kl_divergence(X.T, W, H), rtol=1e-12)
    e.close()


@pytest.mark.parametrize("V,N,K,n_given", [(96, 1500, 70, 3), (96, 700, 130, 66), (96, 900, 100, 100), (50, 300, 97, 0)])
def test_many_signatures_weighted_lhalf_given_and_zeros(V, N, K, n_given):
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=5 * K)
    rng = np.random.default_rng(K)
    wk, wl = rng.uniform(0.5, 2.0, N), rng.uniform(0.0, 0.4, N)
    Xz = X.copy()
    Xz[rng.random(X.shape) < 0.05] = 0.0  # zeros through the function-level API (no clip)
    e = Engine(N, V, K)
    e.upload_X(Xz), e.upload_W(W0), e.upload_H(H0)
    e.set_weights(wk, wl)
    assert np.isclose(e.objective(), orc.klnmf_objective(Xz.T, W0.T, H0.T, wk, wl), rtol=1e-12)
    W, H = W0.T, H0.T
    for _ in range(4):
        W, H = orc.update_WH(Xz.T, W, H, wk, wl, n_given)
    e.kl_step(4, n_given)
    assert rel_l2(e.download_W(), W.T) < 1e-12 and rel_l2(e.download_H(), H.T) < 1e-12
    assert np.array_equal(e.download_W()[:n_given], np.clip(W0[:n_given], orc.EPSILON, None))
    # the kept block / rollback, the queued objective and the objective queued with its following steps
    Wk, Hk = e.download_W(), e.download_H()
    e.objective_async(3)
    e.kl_step_keep(3, n_given)
    W3, H3 = e.download_W(), e.download_H()
    e.kl_rollback()
    assert np.array_equal(e.download_W(), Wk) and np.array_equal(e.download_H(), Hk)
    e.kl_step_objective(4, 3, n_given, keep=True)
    assert np.array_equal(e.download_W(), W3) and np.array_equal(e.download_H(), H3)
    e.kl_rollback()
    vals = e.objective_read(3, 2)
    assert vals[0] == e.objective() and vals[1] == vals[0]
    e.close()


def test_many_signatures_refusals_and_limits():
    N, K = 500, 80
    X, W0, H0 = orc.synthetic_problem(96, N, K, seed=1)
    e = Engine(N, 96, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    for call in (lambda: e.corr_configure(4), lambda: e.set_precision("f32"), lambda: e.set_H_scale(np.ones(K))):
        with pytest.raises(RuntimeError, match="n_signatures > 64"):
            call()
    e.close()
    with pytest.raises(RuntimeError, match="n_signatures"):
        Engine(10, 96, 513)


def test_many_signatures_model_fit_matches_the_oracle_fit():
    """``KLNMF.fit`` with 80 signatures: queued objectives,
    tolerance stop -- same iterations, history and factors as the restated reference loop."""
    V, N, K = 96, 3000, 80
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=11)
    kw = dict(min_iterations=30, max_iterations=2000, conv_test_freq=10, tol=1e-5)
    m = sal.models.KLNMF(K, "custom", **kw)
    m.fit(sal.AnnData(X.copy()), init_kwargs={"signatures_mat": W0.copy(), "exposures_mat": H0.copy()})
    W, H, it, hist = orc.fit_klnmf(X.T, W0.T, H0.T, **kw)
    assert m.n_iterations_ == it and np.allclose(m.history["objective_function"], hist, rtol=1e-11)
    assert rel_l2(m.asignatures.X, W.T) < 1e-8 and rel_l2(m.adata.obsm["exposures"], H.T) < 1e-8
    d = sal.models.KLNMF(K, "nndsvd", min_iterations=5, max_iterations=5)
    d.fit(sal.AnnData(X.copy()), init_kwargs={"seed": 0})
    assert np.all(np.isfinite(d.asignatures.X)) and len(d.history["objective_function"]) == 0


# ------------------------------------------------------------------ MvNMF on more than 64 signatures
@pytest.mark.parametrize("V,N,K,n_given,lam,delta", [(96, 1500, 65, 0, 0.6, 0.8), (96, 2100, 100, 3, 1.0, 1.0), (80, 900, 130, 0, 0.3, 2.0),
                                                     (96, 1300, 200, 0, 50.0, 0.5)])
def test_many_signatures_mvnmf_steps_and_function_level_api_match_the_oracle(V, N, K, n_given, lam, delta):
    """The reference's MvNMF has no limit on ``n_signatures`` (``mvnmf.py:116-126``).  Beyond 64 the engine runs the step in
    its plain form over the signature chunks -- update_H and the numerators on the chunks' ratio passes, the K x K algebra
    (Gram matrix, elimination without pivoting, log det, ``A = Y_minus W``, ``B = |Y| W``) in global memory, a host-driven
    line search whose objectives are the chained forward passes with the trial's column scale applied on the fly: five steps
    against ``orc.mvnmf_step`` (gamma sequence exact), the objective, and every function of the function-level API.  Sample
    weights set on the engine are ignored, as everywhere on the MvNMF path."""
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=V + K)
    e = Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.set_weights(np.random.default_rng(K).uniform(0.5, 2.0, N), None)
    assert np.isclose(e.mv_logdet(delta), orc.volume_logdet(W0.T, delta), rtol=1e-11)
    assert np.isclose(e.mv_objective(lam, delta), orc.kl_divergence_penalized(X.T, W0.T, H0.T, lam, delta), rtol=1e-11)
    Wu = e.mv_update_W_unconstrained(n_given, lam, delta)
    assert rel_l2(Wu, orc.update_W_unconstrained(X.T, W0.T, H0.T, lam, delta, n_given).T) < 1e-7
    W, H, g, gs = W0.T, H0.T, 1.0, []
    for _ in range(5):
        W, H, g = orc.mvnmf_step(X.T, W, H, lam, delta, g, n_given)
        gs.append(g)
    gamma, got = 1.0, []
    for _ in range(2):
        gamma = e.mv_step(1, n_given, lam, delta, gamma)
        got.append(gamma)
    gamma, obj = e.mv_step_objective(3, n_given, lam, delta, gamma)
    assert np.allclose(got + [gamma], [gs[0], gs[1], gs[4]], rtol=1e-12)
    assert rel_l2(e.download_W(), W.T) < 1e-7 and rel_l2(e.download_H(), H.T) < 1e-7
    assert np.isclose(obj, orc.kl_divergence_penalized(X.T, W, H, lam, delta), rtol=1e-9)
    assert np.isclose(e.mv_objective(lam, delta), obj, rtol=1e-9)
    # the two halves of _update_W one by one (MvNMF._update_W_unconstrained / _line_search)
    e.set_weights(None, None)
    e.update_H()
    H1 = orc.update_H(X.T, W, H)
    Wu = e.mv_update_W_unconstrained(n_given, lam, delta)
    g2 = e.mv_line_search(lam, delta, gamma, Wu)
    Wn, Hn, g2_want = orc.line_search(X.T, W, H1, lam, delta, gamma, orc.update_W_unconstrained(X.T, W, H1, lam, delta, n_given))
    assert np.isclose(g2, g2_want, rtol=1e-12) and rel_l2(e.download_W(), Wn.T) < 1e-7 and rel_l2(e.download_H(), Hn.T) < 1e-7
    e.close()


def test_many_signatures_mvnmf_backtracking_and_model_fit():
    """A line search that backtracks on 70 signatures (low counts, a dominant volume penalty), and ``MvNMF(70).fit`` end to
    end: same iterations, history and factors as the restated reference loop."""
    V, N, K = 96, 400, 70
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=77, mean_mutations=100.0)
    X = X.clip(orc.EPSILON)
    lam, delta = 1.0e4, 1.0  # (the oracle's gammas: 1, 1, 1, 0.768, 0.590, 0.566)
    W, H, g, gs = W0.T, H0.T, 1.0, []
    for _ in range(6):
        W, H, g = orc.mvnmf_step(X.T, W, H, lam, delta, g, 0)
        gs.append(g)
    e = Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    gamma, got = 1.0, []
    for _ in range(6):
        gamma = e.mv_step(1, 0, lam, delta, gamma)
        got.append(gamma)
    assert min(gs) < 1.0, gs  # (the case does backtrack)
    assert np.allclose(got, gs, rtol=1e-12), (got, gs)
    assert rel_l2(e.download_W(), W.T) < 1e-7 and rel_l2(e.download_H(), H.T) < 1e-7
    e.close()
    kw = dict(min_iterations=20, max_iterations=40, conv_test_freq=10, tol=1e-6)
    m = sal.models.MvNMF(K, "custom", lam=0.5, delta=1.0, **kw)
    m.fit(sal.AnnData(X.copy()), init_kwargs={"signatures_mat": W0.copy(), "exposures_mat": H0.copy()})
    Wf, Hf, _, it, hist = orc.fit_mvnmf(X.T, W0.T, H0.T, lam=0.5, delta=1.0, **kw)
    assert m.n_iterations_ == it and np.allclose(m.history["objective_function"], hist, rtol=1e-7)
    assert rel_l2(m.asignatures.X, Wf.T) < 1e-6 and rel_l2(m.adata.obsm["exposures"], Hf.T) < 1e-6
