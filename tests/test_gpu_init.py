"""GPU tests of the device-side initialisation: ``salnmf_init_*`` and ``device_init.py``
against NumPy, the reference's initialisation fixtures and scikit-learn's NNDSVD (what the reference calls)."""

import os

import numpy as np
import pytest

import salamander_amd as sal
from conftest import REF_FIX, rel_l2
from oracle import klnmf_oracle as orc
from salamander_amd import initialization as init
from salamander_amd.device_init import initialize_on_device
from salamander_amd.engine import Engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,V,K", [(10, 96, 2), (1000, 96, 50), (333, 83, 7), (4097, 17, 16), (700, 96, 80), (901, 288, 100), (300, 250, 130)])
def test_gram_and_projection_primitives(N, V, K):
    rng = np.random.default_rng(N)
    X = rng.poisson(rng.gamma(1.0, 20.0, size=(N, V))).astype(float)
    e = Engine(N, V, K)
    e.upload_X(X)
    G, total = e.init_gram()
    want = X.T @ X
    assert np.allclose(G, want, rtol=1e-13, atol=1e-9) and np.array_equal(G, G.T)
    assert np.isclose(total, X.sum(), rtol=1e-14)
    B = rng.normal(size=(K, V))
    pos2, neg2 = e.init_project(B)
    U = X @ B.T
    assert rel_l2(e.download_H(), U) < 1e-14
    assert np.allclose(pos2, (np.maximum(U, 0) ** 2).sum(axis=0), rtol=1e-12)
    assert np.allclose(neg2, (np.minimum(U, 0) ** 2).sum(axis=0), rtol=1e-12)
    scale, post = rng.uniform(0.5, 2.0, K), rng.uniform(0.5, 2.0, K)
    take_neg = rng.integers(0, 2, K).astype(np.int32)
    e.init_finish(scale, take_neg, post, 1e-6, 0.25)
    E = np.where(take_neg[None, :].astype(bool), np.maximum(-U, 0), np.maximum(U, 0)) * scale[None, :]
    E[:, 0] = np.abs(U[:, 0]) * scale[0]
    E[E < 1e-6] = 0
    E[E == 0] = 0.25
    assert rel_l2(e.download_H(), np.clip(E * post[None, :], orc.EPSILON, None)) < 1e-14
    e.init_flat(post)
    assert rel_l2(e.download_H(), np.clip((X.sum(axis=1) / K)[:, None] * post[None, :], orc.EPSILON, None)) < 1e-14
    e.upload_W(rng.dirichlet(np.ones(V), size=K))
    e.kl_step(1)  # the padded layout the init kernels leave behind is a valid engine state
    assert np.all(np.isfinite(e.download_H()))
    e.close()


@pytest.mark.parametrize("method", ["flat", "nndsvd", "nndsvda"])
def test_device_init_matches_reference_fixtures(method):
    """The reference's fixtures (tests/test_initialization.py:28-53; 96 x 10 counts, K = 2): entry by entry."""
    d = os.path.join(REF_FIX, "initialization")
    data = np.load(f"{d}/data_mat.npy")
    suffix = "flat.npy" if method == "flat" else f"{method}_seed1.npy"
    e = Engine(data.shape[0], data.shape[1], 2)
    e.upload_X(data)
    S = initialize_on_device(e, 2, method)
    assert np.allclose(S, np.load(f"{d}/signatures_mat_{suffix}"), rtol=1e-7, atol=1e-12)
    assert np.allclose(e.download_H(), np.load(f"{d}/exposures_mat_{suffix}"), rtol=1e-7, atol=1e-12)
    e.close()


def _principal_angles_cos(A, B):
    qa, _ = np.linalg.qr(A.T)
    qb, _ = np.linalg.qr(B.T)
    return np.linalg.svd(qa.T @ qb, compute_uv=False)


@pytest.mark.parametrize("method", ["nndsvd", "nndsvda"])
def test_device_nndsvd_at_c2_against_sklearn(method):
    """Config c2 (96 x 100 000, K = 50): the exact-SVD initialisation on the device against sklearn's randomized-SVD
    NNDSVD on the host (seeded, as the reference would run it).  The leading components -- well separated singular
    values, where the randomized SVD has converged -- agree entry by entry; over all 50 the signatures span the same
    subspace (the trailing singular vectors of a noisy count matrix are determined only up to rotations among
    nearly equal singular values, so entries are not comparable there); and as a starting point the device
    initialisation is as good: the KL objective after 20 steps is within 1 % of the host initialisation's."""
    X, _, _ = orc.synthetic_problem(96, 100000, 50, seed=0)
    S_host, E_host = init.initialize_mat(X, 50, method, seed=1)
    e = Engine(100000, 96, 50)
    e.upload_X(X)
    S_dev = initialize_on_device(e, 50, method)
    E_dev = e.download_H()
    assert np.all(np.isfinite(S_dev)) and np.all(np.isfinite(E_dev)) and S_dev.min() >= orc.EPSILON and E_dev.min() >= orc.EPSILON
    assert np.allclose(S_dev.sum(axis=1), 1.0, atol=1e-4)
    lead = 5
    # (nndsvda: an entry on the zero threshold in one result and just above it in the other differs by the fill value)
    assert rel_l2(S_dev[:lead], S_host[:lead]) < 1e-3 and rel_l2(E_dev[:, :lead], E_host[:, :lead]) < (1e-3 if method == "nndsvd" else 2e-2)
    if method == "nndsvd":  # (the nndsvda fill adds the same constant to both: compare the spans before it)
        cos = _principal_angles_cos(S_dev, S_host)
        # (the last few directions sit at the noise floor of the counts: the randomized SVD has not converged there)
        assert np.median(cos) > 0.999 and (cos > 0.95).sum() >= 44 and cos.min() > 0.5
    objs = []
    for S, E in ((S_dev, E_dev), (S_host, E_host)):
        e.upload_W(np.ascontiguousarray(S)), e.upload_H(np.ascontiguousarray(E))
        e.kl_step(20)
        objs.append(e.objective())
    assert abs(objs[0] - objs[1]) / objs[1] < 0.01
    e.close()


def test_default_fit_initialises_on_the_device(golden):
    """KLNMF(K).fit(adata) with the default init_method runs the device initialisation and a normal fit; with a
    seed it takes the reference's host path and ends close to it (same SVD up to the randomized approximation)."""
    X = golden.pcawg["X"].T.copy()
    fits = []
    for kwargs in (None, {"seed": 1}):
        m = sal.models.KLNMF(5, min_iterations=50, max_iterations=50)
        m.fit(sal.AnnData(X.copy()), init_kwargs=kwargs)
        fits.append(m)
        assert np.all(np.isfinite(m.asignatures.X)) and len(m.history["objective_function"]) == 5
    a, b = (f.history["objective_function"][-1] for f in fits)
    assert abs(a - b) / b < 1e-6
    assert rel_l2(fits[0].asignatures.X, fits[1].asignatures.X) < 1e-5


def test_corrnmf_models_default_init_on_the_device():
    """CorrNMFDet / MultimodalCorrNMF with the default nndsvd: the signatures come from the device initialisation
    (X uploaded once and kept for the fit) and agree with the host path a seed selects (the embeddings are random
    draws either way, so whole fits are not comparable); the fits run and their ELBO increases."""
    from salamander_amd.models import CorrNMFDet, MultimodalCorrNMF

    X1, _, _ = orc.synthetic_problem(96, 3000, 6, seed=3)
    X2, _, _ = orc.synthetic_problem(83, 3000, 4, seed=4)
    sigs = []
    for kwargs in (None, {"seed": 1}):
        m = CorrNMFDet(n_signatures=6, dim_embeddings=3)
        m._setup_adata(sal.AnnData(X1.copy()))
        m._initialize(None, kwargs)
        sigs.append(np.asarray(m.asignatures.X))
        assert (m._resident == {"X"}) == (kwargs is None)
    assert rel_l2(sigs[0], sigs[1]) < 1e-3
    np.random.seed(5)
    m = CorrNMFDet(n_signatures=6, dim_embeddings=3, min_iterations=30, max_iterations=30)
    m.fit(sal.AnnData(X1.copy()))
    hist = m.history["objective_function"]
    assert np.all(np.isfinite(hist)) and hist[-1] > hist[0] and np.allclose(m.asignatures.X.sum(axis=1), 1.0, atol=1e-4)

    sigs = []
    for kwargs in (None, {"seed": 1}):
        mm = MultimodalCorrNMF([6, 4], dim_embeddings=3)
        mm._setup_mdata(sal.MuData({"sbs": sal.AnnData(X1.copy()), "indel": sal.AnnData(X2.copy())}))
        mm._initialize(None, kwargs)
        sigs.append([np.asarray(mm.asignatures[n].X) for n in ("sbs", "indel")])
        assert list(mm.asignatures["indel"].obs_names)[0] == "indel Sig1"
    assert rel_l2(sigs[0][0], sigs[1][0]) < 1e-3 and rel_l2(sigs[0][1], sigs[1][1]) < 1e-3
    np.random.seed(5)
    mm = MultimodalCorrNMF([6, 4], dim_embeddings=3, min_iterations=30, max_iterations=30)
    mm.fit(sal.MuData({"sbs": sal.AnnData(X1.copy()), "indel": sal.AnnData(X2.copy())}))
    hist = mm.history["objective_function"]
    assert np.all(np.isfinite(hist)) and hist[-1] > hist[0]


# ---------------------------------------------------------------- separableNMF: the K deflation rounds on the device
def _host_selection(X, K):
    """The reference's loop (methods.py:112-135), as ``initialization.init_separableNMF`` restates it."""
    R = X.T / X.T.sum(axis=0)
    chosen = []
    for _ in range(K):
        norms = (R**2).sum(axis=0)
        j = int(np.argmax(norms))
        u = R[:, j]
        R = R - np.outer(u, u @ R) / norms[j]
        chosen.append(j)
    return np.array(chosen)


def test_separable_selection_matches_reference_fixture_index_for_index():
    """The reference's separableNMF fixture (tests/test_initialization.py:28-53; seed 1, K = 2): the device selects the
    same samples in the same order, and the initialisation built on its indices equals the fixture."""
    d = os.path.join(REF_FIX, "initialization")
    data = np.load(f"{d}/data_mat.npy")
    e = Engine(data.shape[0], data.shape[1], 2)
    e.upload_X(data)
    chosen = e.init_separable(2)
    e.close()
    assert chosen.tolist() == _host_selection(data, 2).tolist()
    S, E = init.initialize_mat(data, 2, "separableNMF", seed=1, chosen=chosen)
    assert np.allclose(S, np.load(f"{d}/signatures_mat_separableNMF_seed1.npy"))
    assert np.allclose(E, np.load(f"{d}/exposures_mat_separableNMF_seed1.npy"))


@pytest.mark.parametrize("N,V,K", [(3000, 96, 50), (777, 83, 12), (20001, 96, 30), (40, 7, 5)])
def test_separable_selection_equals_the_host_loop(N, V, K):
    """Synthetic counts (incl. ragged N, V < 96, zeros clipped to EPSILON as fit() hands them over): the same K indices
    as the host's deflation loop, whose norms differ from the device's only in summation order."""
    X, _, _ = orc.synthetic_problem(V, N, min(K, 8), seed=N)
    e = Engine(N, V, K)
    e.upload_X(X)
    chosen = e.init_separable(K)
    e.close()
    assert chosen.tolist() == _host_selection(X, K).tolist()
    assert len(set(chosen.tolist())) == K


def test_model_separable_init_uses_the_device_and_equals_the_host_path():
    """``init_method="separableNMF"``: the model takes the indices from the device (X stays resident for the fit) and the
    exposures from the host's seeded RNG -- the same initial state as the reference's host computation, bit for bit."""
    X, _, _ = orc.synthetic_problem(96, 5000, 6, seed=3)
    states = []
    for device_init in (True, False):
        m = sal.models.KLNMF(20, "separableNMF", min_iterations=0, max_iterations=0, device_init=device_init)
        m._setup_adata(sal.AnnData(X.copy()))
        m._initialize(None, {"seed": 7})
        states.append((np.array(m.asignatures.X), np.array(m.adata.obsm["exposures"]), set(m._resident)))
    assert np.array_equal(states[0][0], states[1][0]) and np.array_equal(states[0][1], states[1][1])
    assert states[0][2] == {"X"} and states[1][2] == set()
    m = sal.models.KLNMF(20, "separableNMF", min_iterations=20, max_iterations=20)
    m.fit(sal.AnnData(X.copy()), init_kwargs={"seed": 7})
    W, H, _, hist = orc.fit_klnmf(X.T, states[1][0].T, states[1][1].T, min_iterations=20, max_iterations=20)
    assert rel_l2(m.asignatures.X, W.T) < 1e-10 and np.allclose(m.history["objective_function"], hist, rtol=1e-11)


def test_separable_selection_of_a_rank_deficient_catalogue_keeps_the_host_path():
    """More signatures than the rank of X (duplicated samples): after the first ``rank`` rounds every remaining norm is
    rounding noise, where the device's argmax (fused multiply-adds, another summation order) need not be ``np.argmax``'s.
    The engine reports the winning norms; the model sees them collapse and keeps the reference's host selection, so the
    initial state is the host path's bit for bit."""
    rng = np.random.default_rng(0)
    base = rng.poisson(50.0, size=(3, 96)).astype(float) + 1.0
    X = base[rng.integers(0, 3, size=400)]  # 400 samples, three distinct rows: rank 3
    e = Engine(400, 96, 6)
    e.upload_X(X)
    chosen, norms = e.init_separable(6, return_norms=True)
    e.close()
    assert norms[0] > 0 and np.all(norms[3:] <= 1e-12 * norms[0])  # the collapse is visible
    states = []
    for device_init in (True, False):
        m = sal.models.KLNMF(6, "separableNMF", min_iterations=0, max_iterations=0, device_init=device_init)
        m._setup_adata(sal.AnnData(X.copy()))
        m._initialize(None, {"seed": 1})
        states.append((np.array(m.asignatures.X), np.array(m.adata.obsm["exposures"])))
    assert np.array_equal(states[0][0], states[1][0]) and np.array_equal(states[0][1], states[1][1])


@pytest.mark.parametrize("V,N,K,method", [(96, 1500, 80, "nndsvd"), (288, 1203, 100, "nndsvda"), (192, 800, 70, "flat")])
def test_device_init_on_signature_chunks_equals_numpy(V, N, K, method):
    """More than 64 signatures (round 5): the projection, the NNDSVD post-processing and the flat initialisation run chunk
    by chunk.  Against the same computation in NumPy (exact SVD through the Gram matrix, sklearn's NNDSVD loop on it,
    ``normalize_WH`` + clip): signatures and exposures entry by entry."""
    from salamander_amd.device_init import nndsvd_signature_side

    rng = np.random.default_rng(V + K)
    Wt = rng.dirichlet(np.full(V, 0.3), size=K)
    X = rng.poisson(rng.gamma(0.5, 80.0, size=(N, K)) @ Wt).astype(float).clip(orc.EPSILON)
    e = Engine(N, V, K)
    e.upload_X(X)
    S = initialize_on_device(e, K, method)
    H = e.download_H()
    if method == "flat":
        S_want = np.full((K, V), 1.0 / V).clip(orc.EPSILON)
        H_want = np.repeat(X.sum(axis=1)[:, None] / K, K, axis=1).clip(orc.EPSILON)
    else:
        evals, evecs = np.linalg.eigh(X.T @ X)
        order = np.argsort(evals)[::-1][:K]
        evals, evecs = evals[order], evecs[:, order]
        U = X @ (evecs / np.sqrt(evals))
        pos2, neg2 = (np.maximum(U, 0) ** 2).sum(axis=0), (np.minimum(U, 0) ** 2).sum(axis=0)
        S_raw, scale, take_neg, fill = nndsvd_signature_side(evals, evecs, pos2, neg2, K, X.sum() / (N * V), method)
        E = np.where(take_neg[None, :].astype(bool), np.maximum(-U, 0), np.maximum(U, 0)) * scale[None, :]
        E[:, 0] = np.abs(U[:, 0]) * scale[0]
        E[E < 1e-6] = 0
        if fill:
            E[E == 0] = fill
        colsum = S_raw.sum(axis=1)
        S_want, H_want = (S_raw / colsum[:, None]).clip(orc.EPSILON), (E * colsum[None, :]).clip(orc.EPSILON)
    # (the trailing singular vectors of a noisy count matrix carry rounding-level sign / threshold decisions: compare where
    # both sides made the same ones -- all but a handful of entries)
    assert np.all(np.isfinite(S)) and np.all(np.isfinite(H)) and S.shape == (K, V) and H.shape == (N, K)
    close = np.isclose(H, H_want, rtol=1e-6, atol=1e-9)
    assert close.mean() > 0.999, close.mean()
    assert np.isclose(S, S_want, rtol=1e-6, atol=1e-12).mean() > 0.999
    e.upload_W(np.ascontiguousarray(S))
    e.kl_step(3)  # a valid engine state
    assert np.all(np.isfinite(e.download_H())) and np.isfinite(e.objective())
    e.close()
