"""BASELINE.json's full-size configurations on an MI355X: oracle comparison where the oracle
finishes in seconds, size-independent properties beyond that."""

import numpy as np
import pytest

import salamander_amd as sal
from conftest import rel_l2
from oracle import klnmf_oracle as orc
from salamander_amd import Engine

pytestmark = pytest.mark.gpu

EPS = orc.EPSILON


@pytest.fixture(scope="module")
def c2():
    return orc.synthetic_problem(96, 100000, 50, seed=0)  # config c2: 96 x 100 000, K = 50


def test_c2_steps_vs_oracle(c2):
    X, W0, H0 = c2
    e = Engine(100000, 96, 50)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    assert np.isclose(e.objective(), orc.kl_divergence(X.T, W0.T, H0.T), rtol=1e-12)
    W, H = W0.T, H0.T
    done = 0
    for mark in (1, 10):
        for _ in range(mark - done):
            W, H = orc.update_WH(X.T, W, H)
        e.kl_step(mark - done)
        done = mark
        assert rel_l2(e.download_W(), W.T) < 1e-10 and rel_l2(e.download_H(), H.T) < 1e-10
    assert np.isclose(e.objective(), orc.kl_divergence(X.T, W, H), rtol=1e-12)
    e.close()


def test_c2_north_star_gate_after_100_steps(c2):
    """BASELINE.json's acceptance figure at the headline configuration: "W, H within 1e-4 rel-L2 of the reference after
    equal iterations" (T in {1, 10, 100, 500}; T = 500 is checked by ``bench.py`` against its own CPU run and
    printed with the line: ``time_to_kl.rel_l2_W / rel_l2_H``).  The oracle's ``update_WH`` restates
    ``_utils_klnmf.py:281-361``; 100 of its steps at 96 x 100 000 take ~20 s on the box's host cores.  The north star's
    1e-4 is asserted with the repo's own bar of 1e-9 beside it (fp64 end to end: only summation orders differ)."""
    X, W0, H0 = c2
    e = Engine(100000, 96, 50)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    Xt = np.asfortranarray(X.T)
    W, H = W0.T.copy(), H0.T.copy()
    done, worst = 0, {}
    for mark in (1, 10, 100):
        for _ in range(mark - done):
            W, H = orc.update_WH(Xt, W, H)
        e.kl_step(mark - done)
        done = mark
        worst[mark] = (rel_l2(e.download_W(), W.T), rel_l2(e.download_H(), H.T))
        assert max(worst[mark]) <= 1e-4, worst  # the north star's gate
        assert max(worst[mark]) < 1e-9, worst
    assert np.isclose(e.objective(), orc.kl_divergence(Xt, W, H), rtol=1e-12)
    e.close()


def test_c2_properties_over_100_steps(c2):
    X, W0, H0 = c2
    e = Engine(100000, 96, 50)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    objs = [e.objective()]
    for _ in range(10):
        e.kl_step(10)
        objs.append(e.objective())
    assert all(b <= a * (1 + 1e-13) for a, b in zip(objs[:-1], objs[1:]))  # KL is non-increasing under the updates
    W, H = e.download_W(), e.download_H()
    assert np.isfinite(W).all() and np.isfinite(H).all()
    assert (W >= EPS).all() and (H >= EPS).all()
    assert np.allclose(W.sum(axis=1), 1.0, atol=96 * EPS)  # normalised before the clip: sums in [1, 1 + V*eps]
    # the exposures reproduce each sample's total count (stationarity of the KL updates with unit-sum signatures)
    assert np.allclose(H.sum(axis=1), X.sum(axis=1), rtol=1e-6)
    # a further step from the downloaded state agrees with the oracle started from that state
    Wn, Hn = orc.update_WH(X.T, W.T, H.T)
    e.kl_step(1)
    assert rel_l2(e.download_W(), Wn.T) < 1e-10 and rel_l2(e.download_H(), Hn.T) < 1e-10
    e.close()


def test_c2_model_api_fit_matches_oracle_fit(c2):
    """KLNMF(n_signatures=50).fit(adata) end to end: AnnData in, AnnData out, history cadence."""
    X, W0, H0 = c2
    adata = sal.AnnData(X.copy())
    m = sal.models.KLNMF(50, "custom", min_iterations=20, max_iterations=20)
    m.fit(adata, init_kwargs={"signatures_mat": W0.copy(), "exposures_mat": H0.copy()})
    W, H, it, hist = orc.fit_klnmf(X.T, W0.T, H0.T, min_iterations=20, max_iterations=20)
    assert np.allclose(m.history["objective_function"], hist, rtol=1e-12)
    assert rel_l2(m.asignatures.X, W.T) < 1e-10 and rel_l2(adata.obsm["exposures"], H.T) < 1e-10
    assert m.exposures.shape == (100000, 50) and m.signatures.shape == (50, 96)


def test_c3_shard_size_step_vs_oracle():
    """One rank's block of config c3 (96 x 1 000 000 over 8 GPUs = 125 000 rows)."""
    X, W0, H0 = orc.synthetic_problem(96, 125000, 50, seed=1)
    e = Engine(125000, 96, 50)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    e.kl_step(2)
    W, H = W0.T, H0.T
    for _ in range(2):
        W, H = orc.update_WH(X.T, W, H)
    assert rel_l2(e.download_W(), W.T) < 1e-10 and rel_l2(e.download_H(), H.T) < 1e-10
    e.close()


def test_c4_mvnmf_k30_vs_oracle():
    """Config c4: MvNMF n_signatures=30, 96 x 100 000, lam = delta = 1."""
    X, W0, H0 = orc.synthetic_problem(96, 100000, 30, seed=2)
    e = Engine(100000, 96, 30)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    assert np.isclose(e.mv_objective(1.0, 1.0), orc.kl_divergence_penalized(X.T, W0.T, H0.T, 1.0, 1.0), rtol=1e-12)
    W, H, g = W0.T, H0.T, 1.0
    for _ in range(2):
        W, H, g = orc.mvnmf_step(X.T, W, H, 1.0, 1.0, g)
    gg = e.mv_step(2, 0, 1.0, 1.0, 1.0)
    assert gg == g
    assert rel_l2(e.download_W(), W.T) < 1e-8 and rel_l2(e.download_H(), H.T) < 1e-8
    assert np.isclose(e.mv_objective(1.0, 1.0), orc.kl_divergence_penalized(X.T, W, H, 1.0, 1.0), rtol=1e-10)
    e.close()


def test_c3_full_million_samples_properties():
    """The whole 96 x 1 000 000 matrix on one GPU (768 MB + 400 MB): finite, monotone, normalised."""
    N = 1000000
    rng = np.random.default_rng(0)
    X = np.empty((N, 96))
    H0 = np.empty((N, 50))
    W0 = None
    for s in range(8):  # generated shard by shard as the 8-GPU run does
        Xs, Ws, Hs = orc.synthetic_problem(96, N // 8, 50, seed=100 + s)
        X[s * (N // 8):(s + 1) * (N // 8)] = Xs
        H0[s * (N // 8):(s + 1) * (N // 8)] = Hs
        W0 = Ws if W0 is None else W0
    e = Engine(N, 96, 50)
    e.upload_X(X), e.upload_W(W0), e.upload_H(H0)
    o0 = e.objective()
    e.kl_step(5)
    o1 = e.objective()
    assert np.isfinite(o1) and o1 < o0
    W = e.download_W()
    assert np.allclose(W.sum(axis=1), 1.0, atol=96 * EPS)
    # first 16-sample tiles and the last one against the oracle's H after one more step with this W
    H = e.download_H()
    idx = np.r_[0:64, N - 40:N]
    Hn = orc.update_H(X[idx].T, W.T, H[idx].T)
    e.update_H()
    assert rel_l2(e.download_H()[idx], Hn.T) < 1e-10
    e.close()
