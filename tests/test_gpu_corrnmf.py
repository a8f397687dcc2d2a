"""GPU parity of the CorrNMF dense pieces through the C ABI.

Checked against (i) the reference's own fixtures (``tests/test_corrnmf.py:98-175`` data) and
(ii) the pinned NumPy oracle on larger seeded problems, incl. ragged sizes and every kernel
geometry class.  fp64 throughout; tolerances are relative 1e-12 unless stated.
"""

import numpy as np
import pytest

import salamander_amd as sal
from oracle import corrnmf_oracle as co
from oracle import klnmf_oracle as ko
from salamander_amd import _lib
from salamander_amd.engine import Engine
from test_oracle_corrnmf import load_case

pytestmark = pytest.mark.gpu

RTOL = 1e-12


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def engine_from(X, W, beta, alpha, L, U):
    N, V = X.shape
    K, dim = L.shape
    e = Engine(N, V, K)
    e.upload_X(X)
    e.upload_W(W)
    e.corr_configure(dim)
    e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, beta)
    e.corr_upload(_lib.CORR_SAMPLE_SCALINGS, alpha)
    e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L)
    e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U)
    return e


def synthetic(N, K, dim, V=96, seed=0):
    rng = np.random.default_rng(seed)
    X, W0, _ = ko.synthetic_problem(V, N, K, seed=seed)  # X (N, V), W0 (K, V)
    beta = rng.normal(0.0, 0.3, size=K)
    alpha = np.log(X.sum(axis=1) / K) + rng.normal(0.0, 0.1, size=N)
    L = rng.normal(0.0, 0.5, size=(K, dim))
    U = rng.normal(0.0, 0.5, size=(N, dim))
    return X, W0, beta, alpha, L, U


# ------------------------------------------------------------------ the reference's fixtures


@pytest.fixture(params=[1, 2])
def case(request):
    return load_case(request.param)


def test_fixture_objective(case):
    c = case
    e = engine_from(c["X"], c["W"], c["beta"], c["alpha"], c["L"], c["U"])
    e.corr_compute_exposures()
    llh = e.corr_poisson_llh()
    K, dim = c["L"].shape
    var = c["variance"]
    elbo = llh - 0.5 * dim * K * np.log(2 * np.pi * var) - np.sum(c["L"] ** 2) / (2 * var)
    elbo += -0.5 * dim * c["U"].shape[0] * np.log(2 * np.pi * var) - np.sum(c["U"] ** 2) / (2 * var)
    assert np.allclose(elbo, c["objective"])
    assert abs(elbo - c["objective"]) <= 1e-12 * abs(c["objective"])
    e.close()


def test_fixture_aux_and_signatures(case):
    c = case
    e = engine_from(c["X"], c["W"], c["beta"], c["alpha"], c["L"], c["U"])
    e.corr_compute_exposures()
    assert rel(e.download_H(), co.compute_exposures(c["beta"], c["alpha"], c["L"], c["U"])) < RTOL
    e.corr_compute_aux()
    aux = e.corr_download(_lib.CORR_AUX).T
    assert np.allclose(aux, c["aux"])  # the stored nsigs2 fixture is 1.4e-9 away from its own inputs' aux
    assert rel(aux, co.compute_aux(c["X"], c["W"], co.compute_exposures(c["beta"], c["alpha"], c["L"], c["U"]))) < RTOL
    e.corr_update_signatures(0)
    assert np.allclose(e.download_W(), c["W_updated"])
    e.close()


def test_fixture_scalings(case):
    c = case
    e = engine_from(c["X"], c["W"], c["beta"], c["alpha"], c["L"], c["U"])
    e.corr_upload(_lib.CORR_AUX, c["aux"].T)
    e.corr_update_signature_scalings()
    assert np.allclose(e.corr_download(_lib.CORR_SIGNATURE_SCALINGS), c["beta_updated"])
    e.close()
    e = engine_from(c["X"], c["W"], c["beta"], c["alpha"], c["L"], c["U"])
    e.corr_update_sample_scalings()
    assert np.allclose(e.corr_download(_lib.CORR_SAMPLE_SCALINGS), c["alpha_updated"])
    e.close()


# ------------------------------------------------------------------ oracle, larger shapes


@pytest.mark.parametrize(
    "N,K,dim",
    [(203, 1, 1), (1000, 7, 3), (4099, 16, 16), (2500, 18, 5), (3001, 30, 30), (5000, 50, 50), (777, 50, 2), (1234, 64, 64)],
)
def test_dense_pieces_match_oracle(N, K, dim):
    X, W, beta, alpha, L, U = synthetic(N, K, dim, seed=N + K)
    e = engine_from(X, W, beta, alpha, L, U)
    # the order of CorrNMFDet._update_parameters (corrnmf_det.py:157-169), embeddings held fixed
    e.corr_update_sample_scalings()
    alpha1 = co.update_sample_scalings(X, beta, L, U)
    assert np.max(np.abs(e.corr_download(_lib.CORR_SAMPLE_SCALINGS) - alpha1)) < 1e-12
    e.corr_compute_exposures()
    H = co.compute_exposures(beta, alpha1, L, U)
    assert rel(e.download_H(), H) < RTOL
    e.corr_compute_aux()
    aux = co.compute_aux(X, W, H)
    assert rel(e.corr_download(_lib.CORR_AUX).T, aux) < RTOL
    e.corr_update_signature_scalings()
    beta1 = co.update_signature_scalings(aux, alpha1, L, U)
    assert np.max(np.abs(e.corr_download(_lib.CORR_SIGNATURE_SCALINGS) - beta1)) < 1e-12
    llh = e.corr_poisson_llh()
    want = co.poisson_llh(X.T, W.T, H.T)
    assert abs(llh - want) <= 1e-12 * abs(want)
    n_given = K // 3
    e.corr_update_signatures(n_given)
    assert rel(e.download_W(), ko.update_W(X.T, W.T, H.T, n_given_signatures=n_given).T) < RTOL
    # H must be untouched by the aux pass
    assert rel(e.download_H(), H) < RTOL
    e.close()


def test_aux_pass_leaves_klnmf_path_intact():
    # the aux variant shares fused_kernel with the KL-NMF step: a KL step on the same engine afterwards
    X, W, beta, alpha, L, U = synthetic(1500, 50, 4, seed=9)
    e = engine_from(X, W, beta, alpha, L, U)
    e.corr_compute_exposures()
    H = e.download_H()
    e.corr_compute_aux()
    e.kl_step(1)
    Wn, Hn = ko.update_WH(X.T, W.T, H.T)
    assert rel(e.download_W(), Wn.T) < RTOL and rel(e.download_H(), Hn.T) < RTOL
    e.close()


def test_corr_calls_need_configure():
    e = Engine(64, 96, 3)
    with pytest.raises(RuntimeError, match="corr_configure"):
        e._lib.salnmf_corr_compute_exposures  # symbol exists
        _lib.check(e._lib.salnmf_corr_compute_exposures(e._h))
    e.close()


# ------------------------------------------------------------------ model level: the reference's tests/test_corrnmf.py


def model_from(case):
    import pandas as pd

    import salamander_amd as sal
    from salamander_amd.models import corrnmf_det
    from test_oracle_corrnmf import FIX
    import os

    counts = pd.read_csv(os.path.join(FIX, "counts.csv"), index_col=0).T
    adata = sal.AnnData(counts)
    adata.obs["scalings"] = case["alpha"]
    adata.obsm["embeddings"] = case["U"].copy()
    asignatures = sal.AnnData(case["W"].copy())
    asignatures.var_names = adata.var_names
    asignatures.obs["scalings"] = case["beta"]
    asignatures.obsm["embeddings"] = case["L"].copy()
    K, dim = case["L"].shape
    model = corrnmf_det.CorrNMFDet(n_signatures=K, dim_embeddings=dim)
    model.adata = adata
    model.asignatures = asignatures
    model.compute_exposures()
    model.variance = case["variance"]
    return model


def test_model_objective_function(case):
    assert np.allclose(model_from(case).objective_function(), case["objective"])


class TestUpdatesCorrNMFDet:
    def test_update_signatures(self, case):
        m = model_from(case)
        m.update_signatures()
        assert np.allclose(m.asignatures.X, case["W_updated"])

    def test_update_signature_scalings(self, case):
        m = model_from(case)
        m.update_signature_scalings(case["aux"])
        assert np.allclose(m.asignatures.obs["scalings"].values, case["beta_updated"])

    def test_update_sample_scalings(self, case):
        m = model_from(case)
        m.update_sample_scalings()
        assert np.allclose(m.adata.obs["scalings"].values, case["alpha_updated"])

    def test_update_signature_embeddings(self, case):
        m = model_from(case)
        m.update_signature_embeddings(case["aux"])
        assert np.allclose(m.asignatures.obsm["embeddings"], case["L_updated"])

    def test_update_sample_embeddings(self, case):
        m = model_from(case)
        m.update_sample_embeddings(case["aux"])
        assert np.allclose(m.adata.obsm["embeddings"], case["U_updated"])

    def test_update_variance(self, case):
        m = model_from(case)
        m.update_variance()
        assert np.allclose(m.variance, case["variance_updated"])


def test_model_update_parameters_matches_oracle_steps(case):
    c = case
    m = model_from(c)
    W, beta, alpha, L, U, var = c["W"], c["beta"], c["alpha"], c["L"], c["U"], c["variance"]
    for _ in range(3):
        m._update_parameters()
        W, beta, alpha, L, U, var, H = co.corrnmf_det_step(c["X"], W, beta, alpha, L, U, var)
        # the dense pieces agree to rounding; the embeddings to the solver's tolerance
        assert np.allclose(m.asignatures.X, W, rtol=1e-6, atol=1e-12)
        assert np.allclose(m.adata.obsm["exposures"], H, rtol=1e-6)
        assert np.allclose(m.asignatures.obs["scalings"].values, beta, rtol=1e-6, atol=1e-9)
        assert np.allclose(m.adata.obs["scalings"].values, alpha, rtol=1e-6, atol=1e-9)
        assert np.allclose(m.asignatures.obsm["embeddings"], L, rtol=1e-5, atol=1e-7)
        assert np.allclose(m.adata.obsm["embeddings"], U, rtol=1e-5, atol=1e-7)
        assert np.allclose(m.variance, var, rtol=1e-6)


@pytest.mark.parametrize("n_signatures,dim_embeddings", [(1, 1), (2, 1), (2, 2)])
class TestGivenParametersCorrNMFDet:
    """tests/test_corrnmf.py:178-245: every given parameter survives a (3-iteration) fit unchanged."""

    @pytest.fixture
    def model(self, n_signatures, dim_embeddings):
        from salamander_amd.models import corrnmf_det

        return corrnmf_det.CorrNMFDet(n_signatures=n_signatures, dim_embeddings=dim_embeddings, min_iterations=3, max_iterations=3)

    @pytest.fixture
    def adata(self):
        import os

        import pandas as pd

        import salamander_amd as sal
        from test_oracle_corrnmf import FIX

        return sal.AnnData(pd.read_csv(os.path.join(FIX, "counts.csv"), index_col=0).T)

    def test_given_signatures(self, model, adata):
        for n_given in range(1, model.n_signatures + 1):
            given = adata[:n_given, :].copy()
            given.X = given.X / np.sum(given.X, axis=1, keepdims=True)
            model.fit(adata, given_parameters={"asignatures": given})
            assert np.allclose(given.X, model.asignatures.X[:n_given, :])

    def test_given_signature_scalings(self, model, adata):
        given = np.random.uniform(size=model.n_signatures)
        model.fit(adata, given_parameters={"signature_scalings": given})
        assert np.allclose(given, model.asignatures.obs["scalings"].values)

    def test_given_sample_scalings(self, model, adata):
        given = np.random.uniform(size=adata.n_obs)
        model.fit(adata, given_parameters={"sample_scalings": given})
        assert np.allclose(given, model.adata.obs["scalings"].values)

    def test_given_signature_embeddings(self, model, adata):
        given = np.random.uniform(size=(model.n_signatures, model.dim_embeddings))
        model.fit(adata, given_parameters={"signature_embeddings": given})
        assert np.allclose(given, model.asignatures.obsm["embeddings"])

    def test_given_sample_embeddings(self, model, adata):
        given = np.random.uniform(size=(adata.n_obs, model.dim_embeddings))
        model.fit(adata, given_parameters={"sample_embeddings": given})
        assert np.allclose(given, model.adata.obsm["embeddings"])

    def test_given_variance(self, model, adata):
        model.fit(adata, given_parameters={"variance": 3})
        assert np.allclose(3, model.variance)


def test_fit_history_is_the_elbo_of_the_resident_state():
    """fit() for 20 iterations on a mid-size problem: the recorded ELBO equals the oracle's on the final state."""
    import salamander_amd as sal
    from salamander_amd.models import CorrNMFDet

    rng = np.random.default_rng(5)
    X, _, _ = ko.synthetic_problem(96, 300, 4, seed=5)
    adata = sal.AnnData(X)
    np.random.seed(3)
    model = CorrNMFDet(n_signatures=4, dim_embeddings=2, init_method="random", min_iterations=20, max_iterations=20)
    model.fit(adata, init_kwargs={"seed": 3})
    assert len(model.history["objective_function"]) == 2
    assert model.history["objective_function"][1] > model.history["objective_function"][0]
    want = co.elbo_corrnmf(
        adata.X, model.asignatures.X, adata.obsm["exposures"], model.asignatures.obsm["embeddings"], adata.obsm["embeddings"], model.variance
    )
    assert abs(model.history["objective_function"][-1] - want) <= 1e-10 * abs(want)


# ------------------------------------------------------------------ device Newton-CG vs SciPy


def embedding_problem(N, K, dim, seed):
    rng = np.random.default_rng(seed)
    X, W, _ = ko.synthetic_problem(96, N, K, seed=seed)
    beta = rng.normal(0, 0.3, K)
    L = rng.normal(0, 0.7, (K, dim))
    U = rng.normal(0, 0.7, (N, dim))
    alpha = co.update_sample_scalings(X, beta, L, U)
    aux = co.compute_aux(X, W, co.compute_exposures(beta, alpha, L, U))
    return X, W, beta, alpha, L, U, aux


@pytest.mark.parametrize(
    "N,K,dim,maxiter",
    [(300, 1, 1, 3), (300, 2, 2, 3), (300, 7, 3, 3), (200, 30, 30, 3), (200, 50, 8, 3), (120, 64, 64, 3), (200, 7, 3, 0), (100, 30, 30, 0)],
)
def test_sample_embedding_solves_match_installed_scipy(N, K, dim, maxiter):
    """One device Newton-CG solve per sample against scipy.optimize.minimize(method='Newton-CG') on the
    same problems (the reference's call, ``_utils_corrnmf.py:400-407``), incl. SciPy's status codes.

    The iterates agree to rounding (median error ~1e-15); a rounding-level difference can flip a
    termination test and then shows up at the solver's own tolerance, hence the two-level check."""
    from scipy import optimize

    X, W, beta, alpha, L, U, aux = embedding_problem(N, K, dim, seed=N + K + dim)
    var = 0.8
    e = engine_from(X, W, beta, alpha, L, U)
    e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(aux.T))
    status = e.corr_update_sample_embeddings(var, maxiter, return_status=True)
    got = e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
    e.close()
    want = np.empty_like(U)
    want_status = np.empty(N, dtype=int)
    opts = {"maxiter": maxiter} if maxiter > 0 else {}
    for n in range(N):
        sg = (aux[:, n, None] * L).sum(axis=0)
        res = optimize.minimize(
            fun=lambda x: co.embedding_objective(x, L, alpha[n], beta, var, aux[:, n]),
            x0=U[n].copy(),
            method="Newton-CG",
            jac=lambda x: co.embedding_gradient(x, L, alpha[n], beta, var, sg),
            hess=lambda x: co.embedding_hessian(x, L, alpha[n], beta, var),
            options=opts,
        )
        want[n], want_status[n] = res.x, res.status
    scale = np.maximum(np.abs(want).max(axis=1), 1e-3)
    err = np.abs(got - want).max(axis=1) / scale
    if maxiter > 0:  # the reference's setting: the three iterates agree to rounding
        assert np.median(err) < 1e-12
        assert (err < 1e-8).mean() >= 0.9
    # runs to convergence accumulate rounding differences over many CG solves and stop within xtol of the optimum
    assert err.max() < (2e-4 if maxiter > 0 else 10 * dim * 1e-5)  # SciPy's own stopping tolerance is dim * 1e-5
    # at full convergence the last line search works at rounding level: success vs 'precision loss' may differ
    assert (status == want_status).mean() >= (0.97 if maxiter > 0 else 0.8)


def test_sample_embeddings_tiny_entries_are_pushed_away_from_zero():
    # an embedding whose optimum is ~0 in one coordinate: the EPSILON rule of _utils_corrnmf.py:408-409
    X, W, beta, alpha, L, U, aux = embedding_problem(64, 3, 2, seed=1)
    L[:, 1] = 0.0  # the objective does not depend on coordinate 1 except through the prior -> optimum 0
    U[:, 1] = 1e-9
    e = engine_from(X, W, beta, alpha, L, U)
    e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(aux.T))
    e.corr_update_sample_embeddings(1.0, 3)
    got = e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
    e.close()
    want = co.update_sample_embeddings(aux, L, U, beta, alpha, 1.0)  # SciPy, maxiter=3
    assert np.all((np.abs(got[:, 1]) >= co.EPSILON) | (got[:, 1] == 0.0))
    assert np.allclose(got, want, rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("N,K,dim", [(10, 1, 1), (10, 2, 2), (300, 3, 2), (1000, 7, 3), (2000, 30, 30), (3000, 50, 8), (1500, 64, 64)])
def test_signature_embedding_solves_match_installed_scipy(N, K, dim):
    """One workgroup per signature, every evaluation a pass over all samples; SciPy's default iteration
    limit as in the reference (corrnmf_det.py:103-113).  Run to convergence, so compared at the solver's
    tolerance; in practice the iterates coincide to ~1e-14."""
    X, W, beta, alpha, L, U, aux = embedding_problem(N, K, dim, seed=N + K + dim)
    beta = co.update_signature_scalings(aux, alpha, L, U)
    var = 0.8
    e = engine_from(X, W, beta, alpha, L, U)
    e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(aux.T))
    status = e.corr_update_signature_embeddings(var, 0, return_status=True)
    got = e.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS)
    # the sample side is untouched
    assert np.array_equal(e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS), U)
    e.close()
    want = co.update_signature_embeddings(aux, L, U, beta, alpha, var)
    scale = np.maximum(np.abs(want).max(axis=1), 1e-3)
    err = np.abs(got - want).max(axis=1) / scale
    assert err.max() < 10 * dim * 1e-5
    assert np.median(err) < 1e-6
    assert np.all((status == 0) | (status == 2))  # converged, or stopped at rounding level in the last line search


def test_update_embedding_single_problem_both_layouts():
    """_utils_corrnmf.update_embedding: few terms -> wavefront kernel, many terms -> workgroup kernel."""
    from salamander_amd.models import _utils_corrnmf as uc

    rng = np.random.default_rng(4)
    for n_other, dim, kwargs in [(5, 2, {"options": {"maxiter": 3}}), (40, 6, {}), (500, 3, {})]:
        others = rng.normal(0, 0.6, (n_other, dim))
        scalings_other = rng.normal(0, 0.3, n_other)
        aux_vec = rng.gamma(2.0, 3.0, n_other)
        x0 = rng.normal(0, 0.5, dim)
        got = uc.update_embedding(x0, others, 0.3, scalings_other, 0.9, aux_vec, None, **kwargs)
        want = co.update_embedding(x0, others, 0.3, scalings_other, 0.9, aux_vec, **kwargs)
        assert np.allclose(got, want, rtol=1e-6, atol=1e-9)
    with pytest.raises(TypeError):
        uc.update_embedding(x0, others, 0.3, scalings_other, 0.9, aux_vec, None, tol=1e-3)


def test_resident_fit_equals_per_parameter_updates():
    """fit() (everything resident) against the same updates driven one public method at a time."""
    import salamander_amd as sal
    from salamander_amd.models import CorrNMFDet

    X, _, _ = ko.synthetic_problem(96, 257, 5, seed=11)
    np.random.seed(2)
    a = CorrNMFDet(n_signatures=5, dim_embeddings=3, init_method="random", min_iterations=4, max_iterations=4)
    a.fit(sal.AnnData(X.copy()), init_kwargs={"seed": 2})
    np.random.seed(2)
    b = CorrNMFDet(n_signatures=5, dim_embeddings=3, init_method="random")
    b._setup_adata(sal.AnnData(X.copy()))
    b._initialize(None, {"seed": 2})
    for _ in range(4):
        b.update_sample_scalings()
        b.compute_exposures()
        aux = b._compute_aux()
        b.update_signature_scalings(aux)
        b.update_embeddings(aux)
        b.update_variance()
        b.update_signatures()
    assert np.allclose(a.asignatures.X, b.asignatures.X, rtol=1e-9, atol=1e-14)
    assert np.allclose(a.asignatures.obsm["embeddings"], b.asignatures.obsm["embeddings"], rtol=1e-8, atol=1e-12)
    assert np.allclose(a.adata.obsm["embeddings"], b.adata.obsm["embeddings"], rtol=1e-8, atol=1e-12)
    assert np.allclose(a.adata.obs["scalings"].values, b.adata.obs["scalings"].values, rtol=1e-9, atol=1e-12)
    assert np.allclose(a.variance, b.variance, rtol=1e-10)


# ------------------------------------------------------------------ vectors produced by executing the reference (corr_synth.npz)


@pytest.fixture(params=["a", "b"])
def ref(request):
    from test_oracle_corrnmf import load_corr_synth

    return load_corr_synth(request.param)


def test_device_matches_reference_executed_vectors(ref):
    """Every CorrNMF piece on the device against outputs of the reference's own functions (K = 7 and 12)."""
    r = ref
    var = float(r["var"])
    e = engine_from(r["X"], r["W"], r["beta"], r["alpha"], r["L"], r["U"])
    e.corr_compute_exposures()
    assert rel(e.download_H(), r["H"]) < RTOL
    e.corr_compute_aux()
    assert rel(e.corr_download(_lib.CORR_AUX).T, r["aux"]) < RTOL
    K, dim = r["L"].shape
    prior = -0.5 * dim * K * np.log(2 * np.pi * var) - np.sum(r["L"] ** 2) / (2 * var)
    llh = e.corr_poisson_llh()
    assert np.isclose(llh + prior, float(r["elbo_nopen"]), rtol=1e-12)
    ss_sig, ss_samples = e.corr_embedding_sumsq()
    assert np.isclose(ss_sig, np.sum(r["L"] ** 2), rtol=1e-13) and np.isclose(ss_samples, np.sum(r["U"] ** 2), rtol=1e-13)
    e.corr_update_signature_scalings()
    assert np.max(np.abs(e.corr_download(_lib.CORR_SIGNATURE_SCALINGS) - r["beta_upd"])) < 1e-12
    e.close()
    e = engine_from(r["X"], r["W"], r["beta"], r["alpha"], r["L"], r["U"])
    e.corr_update_sample_scalings()
    assert np.max(np.abs(e.corr_download(_lib.CORR_SAMPLE_SCALINGS) - r["alpha_upd"])) < 1e-12
    e.close()
    # the embedding solves from the stored aux (independent of each other, as in the reference's tests)
    e = engine_from(r["X"], r["W"], r["beta"], r["alpha"], r["L"], r["U"])
    e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(r["aux"].T))
    e.corr_update_signature_embeddings(var, 0)
    assert np.allclose(e.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS), r["L_upd"], rtol=1e-6, atol=1e-9)
    e.close()
    e = engine_from(r["X"], r["W"], r["beta"], r["alpha"], r["L"], r["U"])
    e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(r["aux"].T))
    e.corr_update_sample_embeddings(var, 3)
    assert np.allclose(e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS), r["U_upd"], rtol=1e-6, atol=1e-9)
    e.close()


def test_model_three_updates_match_reference_executed_trajectory(ref):
    import salamander_amd as sal
    from salamander_amd.models import CorrNMFDet

    r = ref
    K, dim = r["L"].shape
    adata = sal.AnnData(r["X"].copy())
    adata.obs["scalings"] = r["alpha"]
    adata.obsm["embeddings"] = r["U"].copy()
    asigs = sal.AnnData(r["W"].copy())
    asigs.obs["scalings"] = r["beta"]
    asigs.obsm["embeddings"] = r["L"].copy()
    m = CorrNMFDet(n_signatures=K, dim_embeddings=dim)
    m.adata, m.asignatures, m.variance = adata, asigs, float(r["var"])
    m.compute_exposures()
    for _ in range(3):
        m._update_parameters()
    assert np.allclose(m.asignatures.X, r["W3"], rtol=1e-6, atol=1e-12)
    assert np.allclose(adata.obsm["exposures"], r["H3"], rtol=1e-6)
    assert np.allclose(asigs.obs["scalings"].values, r["beta3"], rtol=1e-6, atol=1e-9)
    assert np.allclose(adata.obs["scalings"].values, r["alpha3"], rtol=1e-6, atol=1e-9)
    assert np.allclose(asigs.obsm["embeddings"], r["L3"], rtol=1e-5, atol=1e-7)
    assert np.allclose(adata.obsm["embeddings"], r["U3"], rtol=1e-5, atol=1e-7)
    assert np.isclose(m.variance, float(r["var3"]), rtol=1e-6)
    assert np.isclose(m.objective_function(), float(r["elbos"][-1]), rtol=1e-9)


@pytest.mark.parametrize("N,K,dim,V", [(1, 1, 1, 96), (17, 1, 1, 7), (33, 64, 1, 16), (5, 3, 3, 83), (250, 2, 2, 96)])
def test_edge_shapes_full_update_matches_oracle(N, K, dim, V):
    """Degenerate shapes through one full resident update: a single sample, a single signature, V < 96, ragged N."""
    import salamander_amd as sal
    from salamander_amd.models import CorrNMFDet

    rng = np.random.default_rng(N + K + dim + V)
    X, W, _ = ko.synthetic_problem(V, N, K, seed=N + K)
    adata = sal.AnnData(X.copy())
    adata.obs["scalings"] = np.log(X.sum(axis=1) / K)
    adata.obsm["embeddings"] = rng.normal(0, 0.3, (N, dim))
    sigs = sal.AnnData(W.copy())
    sigs.obs["scalings"] = rng.normal(0, 0.1, K)
    sigs.obsm["embeddings"] = rng.normal(0, 0.3, (K, dim))
    want = co.corrnmf_det_step(X, W, sigs.obs["scalings"].values.copy(), adata.obs["scalings"].values.copy(),
                               sigs.obsm["embeddings"].copy(), adata.obsm["embeddings"].copy(), 1.0)
    m = CorrNMFDet(n_signatures=K, dim_embeddings=dim)
    m.adata, m.asignatures = adata, sigs
    m.compute_exposures()
    m._update_parameters()
    assert np.allclose(m.asignatures.X, want[0], rtol=1e-6, atol=1e-12)
    assert np.allclose(adata.obsm["exposures"], want[6], rtol=1e-9)
    assert np.allclose(sigs.obsm["embeddings"], want[3], rtol=1e-4, atol=1e-6)
    assert np.allclose(adata.obsm["embeddings"], want[4], rtol=1e-4, atol=1e-6)
    assert np.isclose(m.variance, want[5], rtol=1e-5)


def test_embedding_solves_terminate_on_non_finite_input():
    """NaN / inf in the inputs must not hang a solve: every loop of the Newton-CG is bounded and NaN fails its tests."""
    X, W, beta, alpha, L, U, aux = embedding_problem(40, 3, 2, seed=3)
    U[5, 0] = np.nan
    aux[1, 7] = np.inf
    e = engine_from(X, W, beta, alpha, L, U)
    e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(aux.T))
    e.corr_update_sample_embeddings(1.0, 3)
    e.corr_update_signature_embeddings(1.0, 0)
    e.sync()
    got = e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
    e.close()
    clean = np.ones(40, dtype=bool)
    clean[[5, 7]] = False
    assert np.isfinite(got[clean]).all()


@pytest.mark.parametrize(
    "N,K,dim",
    [(20000, 12, 8), (33000, 40, 40), (17000, 3, 2), (16500, 64, 64),
     # every shape class of the grouped evaluation kernels (salnmf_corr_lockstep.h): row tiles 1 / 2 / 3, packed tiles 1 .. 5,
     # dim a multiple of 16 (the unpacked form), groups that are not full
     (17000, 7, 20), (17000, 11, 30), (16500, 6, 45), (16500, 9, 36), (17000, 13, 33), (17000, 4, 15), (17000, 5, 16), (16500, 6, 32),
     (16500, 7, 48), (17000, 8, 37), (17000, 11, 38), (2500, 40, 40)],  # (38, 40: the LDS-DMA variant; 2 500: a partial last half)
)
def test_lockstep_signature_solves_agree_with_the_single_kernel_form(N, K, dim):
    """From 2 048 samples on the signature solves advance in lockstep rounds (evaluation over chunks x signatures, the
    solvers replayed from their logs).  Same problems, same solver, the sums of an evaluation in a different order:
    the embeddings agree with the one-workgroup-per-signature kernel far inside the solver's own tolerance and the
    status codes are the same; two of the solves are checked against SciPy as well."""
    rng = np.random.default_rng(N + K)
    X, W, _ = ko.synthetic_problem(96, N, K, seed=K)
    beta, L, U = rng.normal(0, 0.3, K), rng.normal(0, 0.4, (K, dim)), rng.normal(0, 0.4, (N, dim))
    out = []
    for lockstep in (True, False):
        e = Engine(N, 96, K)
        e.set_lockstep(lockstep)
        e.upload_X(X), e.upload_W(W)
        e.corr_configure(dim)
        e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, beta)
        e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L)
        e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U)
        e.corr_update_sample_scalings()
        e.corr_compute_exposures()
        e.corr_compute_aux()
        status = e.corr_update_signature_embeddings(0.8, 0, return_status=True)
        out.append((e.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS), status, e.corr_download(_lib.CORR_SAMPLE_SCALINGS), e.corr_download(_lib.CORR_AUX)))
        e.close()
    (La, sa, alpha, aux), (Lb, sb, _, _) = out
    assert np.array_equal(sa, sb)
    assert np.allclose(La, Lb, rtol=1e-7, atol=1e-10)
    for k in (0, K - 1):
        want = co.update_embedding(L[k], U, beta[k], alpha, 0.8, aux[:, k])
        assert np.allclose(La[k], want, rtol=1e-5, atol=1e-8)


def test_lockstep_sharded_path_world_size_one_equals_unsharded():
    """CorrNMFDet(distributed=True) with enough samples for the lockstep solves: the all-reduce of every round's sums
    (world size 1) must not change a bit against the single-GPU model."""
    import os

    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29543")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        rng = np.random.default_rng(3)
        N, K, dim = 18000, 6, 3
        X, W, _ = ko.synthetic_problem(96, N, K, seed=9)
        res = []
        for distributed in (False, True):
            a = sal.AnnData(X.copy())
            a.obs["scalings"] = np.log(X.sum(axis=1) / K)
            a.obsm["embeddings"] = np.random.default_rng(4).normal(0, 0.3, (N, dim))
            sigs = sal.AnnData(W.copy())
            sigs.obs["scalings"] = np.random.default_rng(5).normal(0, 0.1, K)
            sigs.obsm["embeddings"] = np.random.default_rng(6).normal(0, 0.3, (K, dim))
            m = sal.models.CorrNMFDet(n_signatures=K, dim_embeddings=dim, distributed=distributed)
            m.adata, m.asignatures, m.variance = a, sigs, 0.9
            m._sync_to_device()
            m._device_steps(2, None)
            m._sync_from_device()
            res.append((sigs.obsm["embeddings"].copy(), np.asarray(sigs.X).copy(), a.obsm["embeddings"].copy(), m.variance))
            m._engine.close()
        for x, y in zip(*res):
            assert np.array_equal(x, y)
    finally:
        if created:
            dist.destroy_process_group()


# ------------------------------------------------------------------ batched sample solves (sixteen per wavefront, MFMA)


@pytest.mark.parametrize(
    "N,K,dim,maxiter",
    [(10, 2, 2, 3), (333, 7, 3, 3), (1000, 16, 16, 3), (5000, 30, 30, 3), (3000, 40, 40, 3), (777, 48, 17, 3), (2000, 33, 48, 3), (400, 12, 12, 0)],
)
def test_batched_sample_solves_agree_with_one_wavefront_per_sample(N, K, dim, maxiter):
    """The same solver (its resumable form, ``csrc/salnmf_ncg_machine.h``) on the same problems with the sums in the
    MFMA's order: iterates equal to rounding, a rounding-level flip of a termination test in a few solves at most;
    ragged N (slots without a sample, refills in the middle of a wave's range) and every kernel instantiation."""
    X, W, beta, alpha, L, U, aux = embedding_problem(N, K, dim, seed=N + K + dim)
    out = []
    for batched in (False, True):
        e = engine_from(X, W, beta, alpha, L, U)
        e.set_batched_sample_solves(batched)
        e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(aux.T))
        status = e.corr_update_sample_embeddings(0.8, maxiter, return_status=True)
        out.append((e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS), status))
        e.close()
    (Ua, sa), (Ub, sb) = out
    assert np.isfinite(Ub).all()
    scale = np.maximum(np.abs(Ua).max(axis=1), 1e-3)
    err = np.abs(Ub - Ua).max(axis=1) / scale
    if maxiter > 0:  # (runs to convergence accumulate rounding differences over many CG solves and stop within xtol of the optimum)
        assert np.median(err) < 1e-12
        assert (err < 1e-8).mean() >= 0.97
    # a flipped test changes a truncated solve (maxiter = 3) by a fraction of a Newton step: rare, and both results are
    # iterates of the same descent method -- the objective of every such solve must still have decreased
    # (measured with tests/dev/diag_batched.py, SciPy as the judge: 0.2 % of the solves at K = dim = 40 and 1 % of the runs to
    # convergence differ beyond these bounds, in half of them the batched result is the one SciPy agrees with)
    assert (err < (2e-4 if maxiter > 0 else 10 * dim * 1e-5)).mean() >= (0.995 if maxiter > 0 else 0.97)
    for n in np.flatnonzero(err >= 1e-6)[:20]:
        f0 = co.embedding_objective(U[n], L, alpha[n], beta, 0.8, aux[:, n])
        assert co.embedding_objective(Ub[n], L, alpha[n], beta, 0.8, aux[:, n]) < f0
        assert co.embedding_objective(Ua[n], L, alpha[n], beta, 0.8, aux[:, n]) < f0
    # at full convergence the last line search works at rounding level: success vs 'precision loss' may differ
    assert (sa == sb).mean() >= (0.97 if maxiter > 0 else 0.8)


# ------------------------------------------------------------------ the lockstep kernels against vectors the REFERENCE produced
@pytest.mark.parametrize("lockstep", [True, False])
def test_c5_instantiation_signature_solves_match_reference_executed_vectors(lockstep):
    """Five signatures (one group of the packed evaluation kernel), dim 40 (the LDS-DMA variant, ``ls_eval_packed_kernel``
    with three row tiles), 2 304 samples -- above the 2 048 from which the solves run as lockstep rounds: against the
    reference's ``update_embedding`` results for the same problems (``tests/golden/corr_c5.npz``, made by
    ``make_golden.py --corr-c5``), and the one-workgroup-per-signature kernel beside it.  Measured on an MI355X:
    max |diff| 2.5e-15 / 2.3e-15 (lockstep / single kernel), all five solves converged; asserted at 1e-11."""
    from test_oracle_corrnmf import load_c5_signature_solve

    r, var = load_c5_signature_solve()
    N, dim = r["U"].shape
    K = len(r["beta"])
    e = Engine(N, 96, K)
    e.set_lockstep(lockstep)
    e.corr_configure(dim)
    e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, r["beta"])
    e.corr_upload(_lib.CORR_SAMPLE_SCALINGS, r["alpha"])
    e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, r["L"])
    e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, r["U"])
    e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(r["aux"].T))
    status = e.corr_update_signature_embeddings(var, 0, return_status=True)
    got = e.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS)
    e.close()
    err = np.abs(got - r["L_upd"]).max()
    print(f"c5 signature solves, lockstep={lockstep}: max |diff| {err:.2e}, status {status}")
    assert err < 1e-11
