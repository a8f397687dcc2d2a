"""GPU tests of ``MultimodalCorrNMF`` (SURVEY.md 8f row f1, config c5's model): the reference's
``tests/test_mmcorrnmf.py`` re-run against the device implementation, plus oracle comparisons."""

import os

import numpy as np
import pandas as pd
import pytest

import salamander_amd as sal
from oracle import corrnmf_oracle as co
from oracle import klnmf_oracle as ko
from salamander_amd import _lib
from salamander_amd.engine import Engine
from salamander_amd.models.mmcorrnmf import MultimodalCorrNMF
from test_oracle_corrnmf import MMFIX, load_mm_case

pytestmark = pytest.mark.gpu

NS_SIGNATURES = [2, 3]
DIM_EMBEDDINGS = 2


def make_mdata(c=None):
    adatas = {}
    for m in range(2):
        counts = pd.read_csv(os.path.join(MMFIX, f"model{m}_counts.csv"), index_col=0).T
        adatas[f"mod{m}"] = sal.AnnData(counts)
    mdata = sal.MuData(adatas)
    if c is not None:
        mdata.obsm["embeddings"] = c["U"].copy()
        for m in range(2):
            mdata[f"mod{m}"].obs["scalings"] = c["alphas"][m]
    return mdata


@pytest.fixture
def case():
    return load_mm_case()


@pytest.fixture
def model_init(case):
    c = case
    mdata = make_mdata(c)
    asignatures = {}
    for m in range(2):
        asigs = sal.AnnData(c["Ws"][m].copy())
        asigs.var_names = mdata[f"mod{m}"].var_names
        asigs.obs["scalings"] = c["betas"][m]
        asigs.obsm["embeddings"] = c["Ls"][m].copy()
        asignatures[f"mod{m}"] = asigs
    model = MultimodalCorrNMF(ns_signatures=NS_SIGNATURES, dim_embeddings=DIM_EMBEDDINGS)
    model.mdata = mdata
    model.asignatures = asignatures
    model.compute_exposures()
    model.variance = c["variance"]
    return model


def auxs_of(c):
    return {f"mod{m}": c["auxs"][m] for m in range(2)}


def test_init_signature_names(model_init):
    given = {}
    for name, adata in model_init.mdata.mod.items():
        asigs = sal.AnnData(np.zeros((1, adata.n_vars)))
        asigs.obs_names = ["A"]
        asigs.var_names = adata.var_names
        given[name] = {"asignatures": asigs}
    model_init._initialize(given)
    for name, asigs in model_init.asignatures.items():
        for k, sig_name in enumerate(asigs.obs_names):
            assert sig_name == ("A" if k == 0 else f"{name} Sig{k}")


def test_objective_function(model_init, case):
    assert np.allclose(model_init.objective_function(), case["objective"])


class TestUpdatesMultimodalCorrNMF:
    def test_update_signatures(self, model_init, case):
        model_init.update_signatures()
        for m, (name, sigs) in enumerate(model_init.asignatures.items()):
            assert np.allclose(sigs.X, case["W_updated"][m])

    def test_update_sample_scalings(self, model_init, case):
        model_init.update_sample_scalings()
        for m, (name, adata) in enumerate(model_init.mdata.mod.items()):
            assert np.allclose(adata.obs["scalings"], case["alpha_updated"][m])

    def test_update_signature_scalings(self, model_init, case):
        model_init.update_signature_scalings(auxs_of(case))
        for m, (name, sigs) in enumerate(model_init.asignatures.items()):
            assert np.allclose(sigs.obs["scalings"], case["beta_updated"][m])

    def test_compute_aux(self, model_init, case):
        for m, (name, aux) in enumerate(model_init._compute_auxs().items()):
            assert np.allclose(aux, case["auxs"][m])

    def test_update_signature_embeddings(self, model_init, case):
        model_init.update_signature_embeddings(auxs_of(case))
        for m, (name, asigs) in enumerate(model_init.asignatures.items()):
            assert np.allclose(asigs.obsm["embeddings"], case["L_updated"][m])

    def test_update_sample_embeddings(self, model_init, case):
        model_init.update_sample_embeddings(auxs_of(case))
        assert np.allclose(model_init.mdata.obsm["embeddings"], case["U_updated"])

    def test_update_variance(self, model_init, case):
        model_init.update_variance()
        assert np.allclose(model_init.variance, case["variance_updated"])


def test_update_parameters_matches_oracle_steps(model_init, case):
    c = case
    Ws, betas, alphas, Ls, U, var = c["Ws"], c["betas"], c["alphas"], c["Ls"], c["U"], c["variance"]
    for _ in range(3):
        model_init._update_parameters()
        Ws, betas, alphas, Ls, U, var, Hs = co.mm_step(c["Xs"], Ws, betas, alphas, Ls, U, var)
        for m, name in enumerate(model_init.mod_names):
            assert np.allclose(model_init.asignatures[name].X, Ws[m], rtol=1e-6, atol=1e-12)
            assert np.allclose(model_init.mdata[name].obsm["exposures"], Hs[m], rtol=1e-6)
            assert np.allclose(model_init.asignatures[name].obs["scalings"].values, betas[m], rtol=1e-6, atol=1e-9)
            assert np.allclose(model_init.mdata[name].obs["scalings"].values, alphas[m], rtol=1e-6, atol=1e-9)
            assert np.allclose(model_init.asignatures[name].obsm["embeddings"], Ls[m], rtol=1e-5, atol=1e-7)
        assert np.allclose(model_init.mdata.obsm["embeddings"], U, rtol=1e-5, atol=1e-7)
        assert np.allclose(model_init.variance, var, rtol=1e-6)


@pytest.mark.parametrize("ns_signatures,dim_embeddings", [([1, 2], 1), ([2, 2], 1), ([2, 2], 2)])
class TestGivenParametersMultimodalCorrNMF:
    @pytest.fixture()
    def model(self, ns_signatures, dim_embeddings):
        return MultimodalCorrNMF(ns_signatures=ns_signatures, dim_embeddings=dim_embeddings, max_iterations=3)

    @pytest.fixture()
    def mdata(self):
        return make_mdata()

    def test_given_asignatures(self, model, mdata):
        mod0, mod1 = mdata.mod.keys()
        n_sigs0 = model.ns_signatures[0]
        for n_given in range(1, n_sigs0 + 1):
            given0 = mdata.mod[mod0][:n_given, :].copy()
            given0.X = given0.X.astype(float)
            given0.X = given0.X / np.sum(given0.X, axis=1, keepdims=True)
            given_parameters = {mod0: {"asignatures": given0}}
            model.fit(mdata, given_parameters=given_parameters)
            assert np.allclose(given0.X, model.asignatures[mod0].X[:n_given, :])
            assert not np.allclose(given0.X, model.asignatures[mod1].X[:n_given, :])
            if n_given < n_sigs0:
                other = model.asignatures[mod0].X[n_given:, :].copy()
                model._update_parameters(given_parameters)
                assert not np.allclose(other, model.asignatures[mod0].X[n_given:, :])

    def test_given_signature_scalings(self, model, mdata):
        mod0, mod1 = mdata.mod.keys()
        n_sigs0 = model.ns_signatures[0]
        given = np.random.uniform(size=n_sigs0)
        model.fit(mdata, given_parameters={mod0: {"signature_scalings": given}})
        assert np.allclose(given, model.asignatures[mod0].obs["scalings"])
        assert not np.allclose(given, model.asignatures[mod1].obs["scalings"][:n_sigs0])

    def test_given_signature_embeddings(self, model, mdata):
        mod0, mod1 = mdata.mod.keys()
        n_sigs0 = model.ns_signatures[0]
        given = np.random.uniform(size=(n_sigs0, model.dim_embeddings))
        model.fit(mdata, given_parameters={mod0: {"signature_embeddings": given}})
        assert np.allclose(given, model.asignatures[mod0].obsm["embeddings"])
        assert not np.allclose(given, model.asignatures[mod1].obsm["embeddings"][:n_sigs0, :])

    def test_given_sample_scalings(self, model, mdata):
        mod0, mod1 = mdata.mod.keys()
        given = np.random.uniform(size=mdata.n_obs)
        model.fit(mdata, given_parameters={mod0: {"sample_scalings": given}})
        assert np.allclose(given, model.mdata.mod[mod0].obs["scalings"])
        assert not np.allclose(given, model.mdata.mod[mod1].obs["scalings"])

    def test_given_sample_embeddings(self, model, mdata):
        given = np.random.uniform(size=(mdata.n_obs, model.dim_embeddings))
        model.fit(mdata, given_parameters={"sample_embeddings": given})
        assert np.allclose(given, model.mdata.obsm["embeddings"])

    def test_given_variance(self, model, mdata):
        model.fit(mdata, given_parameters={"variance": 3.0})
        assert np.allclose(3.0, model.variance)


@pytest.mark.parametrize("Ks,dim,N", [([7, 5], 3, 300), ([40, 40], 8, 200), ([64, 64], 16, 64), ([3, 4, 5], 2, 150)])
def test_joint_sample_embedding_solve_matches_installed_scipy(Ks, dim, N):
    """salnmf_corr_update_sample_embeddings_multi on 2-3 modalities incl. more than 64 signatures in total
    (two terms per lane) against the oracle's SciPy solves."""
    rng = np.random.default_rng(sum(Ks) + dim)
    U = rng.normal(0, 0.6, (N, dim))
    Xs, Ws, betas, alphas, Ls, auxs, engines = [], [], [], [], [], [], []
    for m, K in enumerate(Ks):
        X, W, _ = ko.synthetic_problem(96, N, K, seed=10 + m)
        beta, L = rng.normal(0, 0.3, K), rng.normal(0, 0.6, (K, dim))
        alpha = co.update_sample_scalings(X, beta, L, U)
        aux = co.compute_aux(X, W, co.compute_exposures(beta, alpha, L, U))
        e = Engine(N, 96, K)
        e.corr_configure(dim)
        e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, beta)
        e.corr_upload(_lib.CORR_SAMPLE_SCALINGS, alpha)
        e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L)
        e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U)
        e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(aux.T))
        engines.append(e)
        betas.append(beta); alphas.append(alpha); Ls.append(L); auxs.append(aux)
    status = Engine.corr_update_sample_embeddings_multi(engines, 0.9, 3, return_status=True)
    got = [e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS) for e in engines]
    for e in engines:
        e.close()
    for g in got[1:]:
        assert np.array_equal(g, got[0])  # every modality's engine holds the shared result
    want = co.mm_update_sample_embeddings(auxs, Ls, U, betas, alphas, 0.9)
    err = np.abs(got[0] - want).max(axis=1) / np.maximum(np.abs(want).max(axis=1), 1e-3)
    assert np.median(err) < 1e-12 and err.max() < 2e-4
    assert set(np.unique(status)) <= {0, 1, 2}


def test_multi_solve_rejects_mismatched_engines():
    a, b = Engine(32, 96, 3), Engine(48, 96, 3)
    a.corr_configure(2)
    b.corr_configure(2)
    with pytest.raises(RuntimeError, match="differs"):
        Engine.corr_update_sample_embeddings_multi([a, b], 1.0, 3)
    a.close()
    b.close()


def test_c5_like_scale_subset_matches_installed_scipy():
    """Config c5's shape at a tenth of its samples -- (96 + 83) x 20 000, 40 signatures per modality, dim 40,
    80 terms per sample solve (two per lane) -- checked on a random subset against SciPy."""
    rng = np.random.default_rng(5)
    N, Ks, Vs, dim, var = 20000, [40, 40], [96, 83], 40, 0.7
    U = rng.normal(0, 0.3, (N, dim))
    betas, alphas, Ls, auxs, engines = [], [], [], [], []
    for m, (K, V) in enumerate(zip(Ks, Vs)):
        X, W, _ = ko.synthetic_problem(V, N, K, seed=20 + m)
        beta, L = rng.normal(0, 0.3, K), rng.normal(0, 0.3, (K, dim))
        e = Engine(N, V, K)
        e.upload_X(X)
        e.upload_W(W)
        e.corr_configure(dim)
        e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, beta)
        e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L)
        e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U)
        e.corr_update_sample_scalings()
        e.corr_compute_exposures()
        e.corr_compute_aux()
        e.corr_update_signature_scalings()
        engines.append(e)
        betas.append(e.corr_download(_lib.CORR_SIGNATURE_SCALINGS))
        alphas.append(e.corr_download(_lib.CORR_SAMPLE_SCALINGS))
        auxs.append(e.corr_download(_lib.CORR_AUX).T.copy())
        Ls.append(L)
    # signature embeddings of modality 0: two of the 40 solves against SciPy
    engines[0].corr_update_signature_embeddings(var, 0)
    L0 = engines[0].corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS)
    for k in (3, 27):
        want = co.update_embedding(Ls[0][k], U, betas[0][k], alphas[0], var, auxs[0][k])
        assert np.allclose(L0[k], want, rtol=1e-5, atol=1e-8)
    # joint sample solve over both modalities (with modality 0's new signature embeddings), 120 samples against SciPy
    Ls[0] = L0
    Engine.corr_update_sample_embeddings_multi(engines, var, 3)
    got = engines[1].corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
    L_all, beta_all, aux_all = np.concatenate(Ls), np.concatenate(betas), np.concatenate(auxs)
    idx = rng.choice(N, size=120, replace=False)
    worst = 0.0
    for n in idx:
        scalings = np.concatenate([np.repeat(alphas[m][n], Ks[m]) for m in range(2)])
        want = co.update_embedding(U[n], L_all, scalings, beta_all, var, aux_all[:, n], options={"maxiter": 3})
        worst = max(worst, np.abs(got[n] - want).max() / max(np.abs(want).max(), 1e-3))
    assert worst < 2e-4
    for e in engines:
        e.close()


def test_joint_solve_matches_reference_executed_vectors():
    from test_oracle_corrnmf import load_mm_synth

    mods, U, U_upd, var = load_mm_synth()
    engines = []
    for m in mods:
        N, V = m["X"].shape
        e = Engine(N, V, len(m["beta"]))
        e.corr_configure(U.shape[1])
        e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, m["beta"])
        e.corr_upload(_lib.CORR_SAMPLE_SCALINGS, m["alpha"])
        e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, m["L"])
        e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U)
        e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(m["aux"].T))
        engines.append(e)
    Engine.corr_update_sample_embeddings_multi(engines, var, 3)
    got = engines[0].corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
    for e in engines:
        e.close()
    assert np.allclose(got, U_upd, rtol=1e-6, atol=1e-9)


# ---------------------------------------------------------------------------------------------------------------
# Config c5 at its real size, and its multi-GPU split (sample-sharded engines)


def _c5_engines(N, rng, var, Ks=(40, 40), Vs=(96, 83), dim=40, rows=None):
    """Engines of the two modalities of a c5-shaped problem (rows = (lo, hi): only that shard of the samples)."""
    lo, hi = rows if rows is not None else (0, N)
    U = rng.normal(0, 0.3, (N, dim))
    engines, state = [], []
    for m, (K, V) in enumerate(zip(Ks, Vs)):
        X, W, _ = ko.synthetic_problem(V, N, K, seed=20 + m)
        beta, L = rng.normal(0, 0.3, K), rng.normal(0, 0.3, (K, dim))
        e = Engine(hi - lo, V, K)
        e.upload_X(X[lo:hi])
        e.upload_W(W)
        e.corr_configure(dim)
        e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, beta)
        e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L)
        e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U[lo:hi])
        engines.append(e)
        state.append(dict(X=X, W=W, beta=beta, L=L))
    return engines, state, U


def _mm_elbo_resident(engines, var, dim, n_total):
    """The ELBO of the resident state, as MultimodalCorrNMF._device_objective computes it."""
    log_norm = np.log(2 * np.pi * var)
    value = 0.0
    for e in engines:
        value += e.corr_poisson_llh()
        value -= 0.5 * dim * e.K * log_norm + e.corr_embedding_sumsq()[0] / (2 * var)
    value -= 0.5 * dim * n_total * log_norm + engines[0].corr_embedding_sumsq()[1] / (2 * var)
    return value


def _mm_update(engines, var):
    for e in engines:
        e.corr_update_sample_scalings()
        e.corr_compute_exposures()
        e.corr_compute_aux()
        e.corr_update_signature_scalings()
    for e in engines:
        e.corr_update_signature_embeddings(var, 0)
    Engine.corr_update_sample_embeddings_multi(engines, var, 3)
    ss = [e.corr_embedding_sumsq() for e in engines]
    count = (sum(e.K for e in engines) + engines[0].N) * engines[0].dim
    var = float(np.clip((sum(s[0] for s in ss) + ss[0][1]) / count, 1e-7, None))
    for e in engines:
        e.corr_update_signatures(0)
    return var


@pytest.mark.parametrize("N", [200000, 50000])
def test_c5_full_size_update_properties_and_subset_vs_oracle(N):
    """Config c5 itself -- (96 + 83) x 200 000, ns_signatures = [40, 40], dim 40 -- and its per-GPU share on 4 GPUs
    (50 000 samples): three full MultimodalCorrNMF updates on one GPU.  Size-independent properties (everything
    finite, the ELBO increases update over update, unit-sum signatures) and a random subset of the signature and
    sample solves of the first update against the oracle (the *installed* SciPy's Newton-CG)."""
    rng = np.random.default_rng(5)
    var, dim = 0.7, 40
    engines, state, U = _c5_engines(N, rng, var)
    # ---- first update piece by piece, keeping what the subset check needs
    for e in engines:
        e.corr_update_sample_scalings()
        e.corr_compute_exposures()
        e.corr_compute_aux()
        e.corr_update_signature_scalings()
    betas = [e.corr_download(_lib.CORR_SIGNATURE_SCALINGS) for e in engines]
    alphas = [e.corr_download(_lib.CORR_SAMPLE_SCALINGS) for e in engines]
    auxs = [e.corr_download(_lib.CORR_AUX).T.copy() for e in engines]
    L_old = [s["L"] for s in state]
    for e in engines:
        e.corr_update_signature_embeddings(var, 0)
    Ls = [e.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS) for e in engines]
    for m, k in ((0, 3), (1, 27)):  # one solve per modality against SciPy (each is a pass over all N samples per evaluation)
        want = co.update_embedding(L_old[m][k], U, betas[m][k], alphas[m], var, auxs[m][k])
        assert np.allclose(Ls[m][k], want, rtol=1e-5, atol=1e-8)
    Engine.corr_update_sample_embeddings_multi(engines, var, 3)
    got = engines[1].corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
    L_all, beta_all, aux_all = np.concatenate(Ls), np.concatenate(betas), np.concatenate(auxs)
    worst = 0.0
    for n in rng.choice(N, size=60, replace=False):
        scalings = np.concatenate([np.repeat(alphas[m][n], 40) for m in range(2)])
        want = co.update_embedding(U[n], L_all, scalings, beta_all, var, aux_all[:, n], options={"maxiter": 3})
        worst = max(worst, np.abs(got[n] - want).max() / max(np.abs(want).max(), 1e-3))
    assert worst < 2e-4
    del auxs, aux_all
    ss = [e.corr_embedding_sumsq() for e in engines]
    var = float(np.clip((sum(s[0] for s in ss) + ss[0][1]) / ((80 + N) * dim), 1e-7, None))
    for e in engines:
        e.corr_update_signatures(0)
    # ---- properties over two more full updates
    elbos = [_mm_elbo_resident(engines, var, dim, N)]
    for _ in range(2):
        var = _mm_update(engines, var)
        elbos.append(_mm_elbo_resident(engines, var, dim, N))
    assert np.all(np.isfinite(elbos)) and elbos[1] > elbos[0] and elbos[2] > elbos[1], elbos
    for e in engines:
        W = e.download_W()
        assert np.all(np.isfinite(W)) and np.allclose(W.sum(axis=1), 1.0, atol=1e-4)
        assert np.all(np.isfinite(e.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS)))
    Ufin = engines[0].corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
    assert np.all(np.isfinite(Ufin)) and np.array_equal(Ufin, engines[1].corr_download(_lib.CORR_SAMPLE_EMBEDDINGS))
    assert 1e-7 <= var < 10
    for e in engines:
        e.close()


def test_sharded_signature_solves_equal_unsharded_bitwise():
    """The exchange point of a sample-sharded CorrNMF update: two engines with half of the samples each, the
    sample-side inputs of the signature solves gathered (here by the host, as a torch.distributed backend would)
    and handed to ``corr_update_signature_embeddings_from`` -- every shard's engine ends with the SAME BITS as the
    engine that holds all samples, because the solves see the same values in the same sample order."""
    rng = np.random.default_rng(11)
    N, K, V, dim, var = 6000, 12, 96, 8, 0.8
    X, W, _ = ko.synthetic_problem(V, N, K, seed=4)
    beta, L, U = rng.normal(0, 0.3, K), rng.normal(0, 0.4, (K, dim)), rng.normal(0, 0.4, (N, dim))

    def prepared(lo, hi):
        e = Engine(hi - lo, V, K)
        e.upload_X(X[lo:hi]), e.upload_W(W)
        e.corr_configure(dim)
        e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, beta)
        e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L)
        e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U[lo:hi])
        e.corr_update_sample_scalings()
        e.corr_compute_exposures()
        e.corr_compute_aux()
        return e

    whole = prepared(0, N)
    status_whole = whole.corr_update_signature_embeddings(var, 0, return_status=True)
    want = whole.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS)
    shards = [prepared(0, 2992), prepared(2992, N)]  # a 16-aligned split, as shard_bounds produces
    gathered = [np.concatenate([e.corr_download(which) for e in shards]) for which in (_lib.CORR_SAMPLE_EMBEDDINGS, _lib.CORR_SAMPLE_SCALINGS, _lib.CORR_AUX)]
    for e in shards:
        status = e.corr_update_signature_embeddings_from(*gathered, var, 0, return_status=True)
        assert np.array_equal(e.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS), want)
        assert np.array_equal(status, status_whole)
        e.close()
    whole.close()


def test_distributed_multimodal_model_world_size_one():
    """MultimodalCorrNMF(distributed=True) on a one-rank process group: communicator per modality, broadcasts, the
    RCCL gather of U / alpha / aux in front of the signature solves, the all-reduced sums -- the same bits as the
    single-GPU model."""
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        results = []
        for distributed in (False, True):
            c = load_mm_case()
            m = MultimodalCorrNMF(NS_SIGNATURES, DIM_EMBEDDINGS, distributed=distributed)
            m.mdata = make_mdata(c)
            m.asignatures = {}
            for i in range(2):
                s = sal.AnnData(c["Ws"][i].copy())
                s.var_names = m.mdata[f"mod{i}"].var_names
                s.obs["scalings"] = c["betas"][i].copy()
                s.obsm["embeddings"] = c["Ls"][i].copy()
                m.asignatures[f"mod{i}"] = s
            m.variance = float(c["variance"])
            m.compute_exposures()
            m._sync_to_device()
            m._device_steps(3, None)
            elbo = m._device_objective()
            m._sync_from_device()
            results.append([m.asignatures[f"mod{i}"].X.copy() for i in range(2)] + [m.asignatures[f"mod{i}"].obsm["embeddings"].copy() for i in range(2)]
                           + [np.asarray(m.mdata.obsm["embeddings"]).copy(), np.array([m.variance, elbo])])
            for e in m._engines.values():
                e.close()
        for a, b in zip(*results):
            assert np.array_equal(a, b)
    finally:
        if created:
            dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------
# The instantiations config c5 uses, against vectors the REFERENCE produced (tests/golden/corr_c5.npz: make_golden.py
# --corr-c5 executes _utils_corrnmf.update_embedding) -- not against the box's SciPy
def _c5_vector_engines(mods, U):
    engines = []
    for m, V in zip(mods, (96, 83)):
        e = Engine(U.shape[0], V, len(m["beta"]))
        e.corr_configure(U.shape[1])
        e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, m["beta"])
        e.corr_upload(_lib.CORR_SAMPLE_SCALINGS, m["alpha"])
        e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, m["L"])
        e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U)
        e.corr_upload(_lib.CORR_AUX, np.ascontiguousarray(m["aux"].T))
        engines.append(e)
    return engines


@pytest.mark.parametrize("batched", [True, False])
@pytest.mark.parametrize("it", [1, 3])
def test_c5_instantiation_joint_sample_solves_match_reference_executed_vectors(it, batched):
    """ns_signatures [40, 40], dim 40, 80 terms per solve, maxiter 3 (``mmcorrnmf.py:398-428``): the batched kernel
    (sixteen solves per wavefront, the instantiation 80 terms x dim 40 that c5 runs) and the one-wavefront-per-sample kernel
    against the reference's own results for 256 samples, at the random start (update 1) and two updates later (update 3).
    Measured on an MI355X (gpurun_out/r05/t_c5.log): max |diff| / max(|row|, 1e-3) over the 256 solves = 7e-15 / 9e-15
    (update 1, batched / per sample) and 3e-14 / 1e-13 (update 3), medians 2-3e-15, every solve ending as the reference's did
    (three Newton iterations) -- asserted at 1e-11, seven orders below the 2e-4 the SciPy-on-the-box comparisons allow."""
    from test_oracle_corrnmf import load_c5_sample_solve

    mods, U, U_upd, var = load_c5_sample_solve(it)
    engines = _c5_vector_engines(mods, U)
    for e in engines:
        e.set_batched_sample_solves(batched)
    status = Engine.corr_update_sample_embeddings_multi(engines, var, 3, return_status=True)
    got = [e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS) for e in engines]
    for e in engines:
        e.close()
    assert np.array_equal(got[0], got[1])
    err = np.abs(got[0] - U_upd).max(axis=1) / np.maximum(np.abs(U_upd).max(axis=1), 1e-3)
    print(f"c5 sample solves, update {it}, batched={batched}: median {np.median(err):.2e} max {err.max():.2e} status {np.unique(status, return_counts=True)}")
    assert err.max() < 1e-11


def test_c5_shape_model_three_updates_match_the_reference_executed_trajectory():
    """``MultimodalCorrNMF._update_parameters`` three times at c5's shape (256 samples) through the device-resident update
    against the trajectory the reference's functions produced from the same start."""
    from test_oracle_corrnmf import load_c5_trajectory

    s, end = load_c5_trajectory()
    N = s["U"].shape[0]
    mdata = sal.MuData({f"mod{m}": sal.AnnData(s["Xs"][m].copy()) for m in range(2)})
    mdata.obsm["embeddings"] = s["U"].copy()
    asignatures = {}
    for m in range(2):
        # (any finite start: the update recomputes the sample scalings first, mmcorrnmf.py:447)
        mdata[f"mod{m}"].obs["scalings"] = np.zeros(N)
        asigs = sal.AnnData(s["Ws"][m].copy())
        asigs.var_names = mdata[f"mod{m}"].var_names
        asigs.obs["scalings"] = s["betas"][m]
        asigs.obsm["embeddings"] = s["Ls"][m].copy()
        asignatures[f"mod{m}"] = asigs
    model = MultimodalCorrNMF(ns_signatures=[40, 40], dim_embeddings=40)
    model.mdata, model.asignatures, model.variance = mdata, asignatures, s["var"]
    model.compute_exposures()
    for _ in range(3):
        model._update_parameters()
    for m, name in enumerate(model.mod_names):
        assert np.allclose(model.asignatures[name].X, end["Ws"][m], rtol=1e-6, atol=1e-12)
        assert np.allclose(model.mdata[name].obsm["exposures"], end["Hs"][m], rtol=1e-6)
        assert np.allclose(model.asignatures[name].obs["scalings"].values, end["betas"][m], rtol=0, atol=1e-7)
        assert np.allclose(model.mdata[name].obs["scalings"].values, end["alphas"][m], rtol=0, atol=1e-7)
        assert np.allclose(model.asignatures[name].obsm["embeddings"], end["Ls"][m], rtol=1e-5, atol=1e-7)
    assert np.allclose(model.mdata.obsm["embeddings"], end["U"], rtol=1e-5, atol=1e-7)
    assert np.isclose(model.variance, end["var"], rtol=1e-7)
