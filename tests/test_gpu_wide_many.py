"""GPU parity for more than 96 features TOGETHER with more than 64 signatures (round 5; VERDICT r4 "what's missing" 3).

The reference has no limit on either (``_utils_klnmf.py:281-361``).  The engine runs such problems per (chunk of <= 64
signatures, block of 96 features): per feature block a chain of forward launches over the chunks forms that block's ratio
X_b / (H W_b); on it every chunk runs its numerator pass and its share of the update_H product, which is accumulated over
the blocks (``salnmf.hip: grid_passes``).  Every KLNMF entry point against the oracle on ragged sizes, with weights, l-half
penalties, given signatures (inside and across chunks) and zeros in X."""

import numpy as np
import pytest

import salamander_amd as sal
from conftest import rel_l2
from oracle import klnmf_oracle as orc
from salamander_amd import Engine, _lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V,N,K", [(97, 500, 65), (288, 1203, 100), (250, 777, 130), (192, 2100, 70), (1536, 300, 80)])
def test_wide_many_functions_against_the_oracle(V, N, K):
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


@pytest.mark.parametrize("V,N,K,n_given", [(288, 900, 70, 3), (200, 700, 130, 66), (192, 500, 100, 100), (150, 300, 97, 0)])
def test_wide_many_weighted_lhalf_given_and_zeros(V, N, K, n_given):
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


@pytest.mark.parametrize("V,N,K,n_given,lam,delta", [(192, 900, 65, 0, 0.6, 0.8), (288, 1100, 100, 3, 1.0, 1.0), (150, 700, 130, 0, 0.3, 2.0)])
def test_wide_many_mvnmf_steps_and_function_level_api_match_the_oracle(V, N, K, n_given, lam, delta):
    """MvNMF (``mvnmf.py:116-126, 190-210``: no limit on either size) on feature blocks and signature chunks together: the plain
    form of the step -- update_H and the numerators over the (chunk, block) pairs, the K x K algebra in global memory, the
    host-driven line search on the chained forward passes -- five steps against ``orc.mvnmf_step`` (gamma sequence exact), the
    objective, and the function-level API."""
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=V + K)
    e = Engine(N, V, K)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.set_weights(np.random.default_rng(K).uniform(0.5, 2.0, N), None)  # ignored on the MvNMF path
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
    e.set_weights(None, None)
    e.update_H()
    H1 = orc.update_H(X.T, W, H)
    Wu = e.mv_update_W_unconstrained(n_given, lam, delta)
    g2 = e.mv_line_search(lam, delta, gamma, Wu)
    Wn, Hn, g2_want = orc.line_search(X.T, W, H1, lam, delta, gamma, orc.update_W_unconstrained(X.T, W, H1, lam, delta, n_given))
    assert np.isclose(g2, g2_want, rtol=1e-12) and rel_l2(e.download_W(), Wn.T) < 1e-7 and rel_l2(e.download_H(), Hn.T) < 1e-7
    e.close()


def test_wide_many_mvnmf_backtracking():
    """A line search that backtracks on 70 signatures x 192 features (low counts, a dominant volume penalty)."""
    V, N, K = 192, 400, 70
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=77, mean_mutations=100.0)
    X = X.clip(orc.EPSILON)
    lam, delta = 1.0e4, 1.0
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


def test_wide_many_model_fits_match_the_oracle_fits():
    """``KLNMF.fit`` and ``MvNMF.fit`` on an SBS-192-sized catalogue with 70 signatures: queued objectives, tolerance stop -- same
    iterations, history and factors as the restated reference loops; the default initialisation (NNDSVD) runs on the device,
    chunk by chunk over the feature blocks, and starts a normal fit."""
    V, N, K = 192, 1500, 70
    X, W0, H0 = orc.synthetic_problem(V, N, K, seed=13)
    kw = dict(min_iterations=30, max_iterations=600, conv_test_freq=10, tol=1e-5)
    m = sal.models.KLNMF(K, "custom", **kw)
    m.fit(sal.AnnData(X.copy()), init_kwargs={"signatures_mat": W0.copy(), "exposures_mat": H0.copy()})
    W, H, it, hist = orc.fit_klnmf(X.T, W0.T, H0.T, **kw)
    assert m.n_iterations_ == it and np.allclose(m.history["objective_function"], hist, rtol=1e-11)
    assert rel_l2(m.asignatures.X, W.T) < 1e-8 and rel_l2(m.adata.obsm["exposures"], H.T) < 1e-8
    kw = dict(min_iterations=20, max_iterations=40, conv_test_freq=10, tol=1e-6)
    mv = sal.models.MvNMF(K, "custom", lam=0.5, delta=1.0, **kw)
    mv.fit(sal.AnnData(X.copy()), init_kwargs={"signatures_mat": W0.copy(), "exposures_mat": H0.copy()})
    Wf, Hf, _, it, hist = orc.fit_mvnmf(X.T, W0.T, H0.T, lam=0.5, delta=1.0, **kw)
    assert mv.n_iterations_ == it and np.allclose(mv.history["objective_function"], hist, rtol=1e-7)
    assert rel_l2(mv.asignatures.X, Wf.T) < 1e-6 and rel_l2(mv.adata.obsm["exposures"], Hf.T) < 1e-6
    d = sal.models.KLNMF(K, min_iterations=20, max_iterations=20)
    d.fit(sal.AnnData(X.copy()))
    h = d.history["objective_function"]
    assert np.all(np.isfinite(d.asignatures.X)) and len(h) >= 2 and h[-1] < h[0]
    assert np.allclose(np.asarray(d.asignatures.X).sum(axis=1), 1.0, atol=1e-6)


def test_wide_many_refusals():
    e = Engine(300, 192, 80)
    for call in (lambda: e.corr_configure(4), lambda: e.set_precision("f32")):
        with pytest.raises(RuntimeError, match="n_signatures > 64|n_features > 96"):
            call()
    e.close()
