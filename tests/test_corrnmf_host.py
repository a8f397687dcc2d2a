"""CPU tests of the CorrNMF host logic that needs no engine: initialisation and validation
(the reference's ``initialize_corrnmf``, ``initialize.py:258-384``)."""

import numpy as np
import pytest

import salamander_amd as sal
from salamander_amd.initialization import initialize_corrnmf
from salamander_amd.models import CorrNMFDet


@pytest.fixture
def adata():
    rng = np.random.default_rng(0)
    return sal.AnnData(rng.poisson(20.0, size=(12, 96)).astype(float))


def test_initialize_corrnmf_defaults(adata):
    np.random.seed(1)
    asigs, variance = initialize_corrnmf(adata, 3, 2, "random", seed=1)
    assert asigs.X.shape == (3, 96) and np.allclose(asigs.X.sum(axis=1), 1.0)
    assert list(asigs.obs_names) == ["Sig1", "Sig2", "Sig3"]
    assert np.array_equal(asigs.obs["scalings"].values, np.zeros(3))
    assert np.array_equal(adata.obs["scalings"].values, np.zeros(12))
    assert asigs.obsm["embeddings"].shape == (3, 2) and adata.obsm["embeddings"].shape == (12, 2)
    assert variance == 1.0
    assert "exposures" not in adata.obsm  # initialize_base leaves adata alone (initialize.py:166-167)


def test_initialize_corrnmf_embeddings_follow_the_global_rng(adata):
    np.random.seed(7)
    initialize_corrnmf(adata, 2, 2, "flat")
    first = adata.obsm["embeddings"].copy()
    np.random.seed(7)
    expected_sig = np.random.multivariate_normal(np.zeros(2), np.identity(2), size=2)
    expected_samples = np.random.multivariate_normal(np.zeros(2), np.identity(2), size=12)
    np.random.seed(7)
    asigs, _ = initialize_corrnmf(adata, 2, 2, "flat")
    assert np.array_equal(asigs.obsm["embeddings"], expected_sig)
    assert np.array_equal(adata.obsm["embeddings"], expected_samples)
    assert np.array_equal(first, expected_samples)


def test_initialize_corrnmf_given_parameters(adata):
    given = {
        "signature_scalings": np.arange(2.0),
        "sample_scalings": np.arange(12.0),
        "signature_embeddings": np.ones((2, 3)),
        "sample_embeddings": np.full((12, 3), 2.0),
        "variance": 3,
    }
    asigs, variance = initialize_corrnmf(adata, 2, 3, "flat", given)
    assert np.array_equal(asigs.obs["scalings"].values, given["signature_scalings"])
    assert np.array_equal(adata.obs["scalings"].values, given["sample_scalings"])
    assert np.array_equal(asigs.obsm["embeddings"], given["signature_embeddings"])
    assert np.array_equal(adata.obsm["embeddings"], given["sample_embeddings"])
    assert variance == 3.0 and isinstance(variance, float)


def test_initialize_corrnmf_without_sample_embeddings(adata):
    initialize_corrnmf(adata, 2, 2, "flat", initialize_sample_embeddings=False)
    assert "embeddings" not in adata.obsm


@pytest.mark.parametrize(
    "given,exc",
    [
        ({"unknown": 1}, ValueError),
        ({"signature_scalings": [0.0, 0.0]}, TypeError),
        ({"signature_scalings": np.zeros(3)}, ValueError),
        ({"sample_scalings": np.zeros(5)}, ValueError),
        ({"signature_embeddings": np.zeros((2, 5))}, ValueError),
        ({"sample_embeddings": np.zeros((12, 1))}, ValueError),
        ({"variance": "1"}, TypeError),
        ({"variance": 0.0}, ValueError),
        ({"variance": -1}, ValueError),
    ],
)
def test_initialize_corrnmf_rejects(adata, given, exc):
    with pytest.raises(exc):
        initialize_corrnmf(adata, 2, 2, "flat", given)


def test_custom_init_is_not_supported(adata):
    with pytest.raises(ValueError, match="Custom"):
        initialize_corrnmf(adata, 2, 2, "custom")


def test_constructor_defaults():
    m = CorrNMFDet(n_signatures=5)
    assert m.dim_embeddings == 5 and m.variance == 1.0 and m.objective == "maximize"
    assert CorrNMFDet(n_signatures=5, dim_embeddings=2).dim_embeddings == 2
    assert (m.min_iterations, m.max_iterations, m.conv_test_freq, m.tol) == (500, 10000, 10, 1e-7)


def test_correlation_of_embeddings(adata):
    np.random.seed(0)
    m = CorrNMFDet(n_signatures=3, dim_embeddings=2, init_method="flat")
    m.adata = adata
    m.asignatures, _ = initialize_corrnmf(adata, 3, 2, "flat")
    m.compute_correlation_scaled("signatures")
    C = m.asignatures.obsp["correlation"]
    L = m.asignatures.obsm["embeddings"]
    assert C.shape == (3, 3) and np.allclose(np.diag(C), 1.0)
    assert np.isclose(C[0, 1], L[0] @ L[1] / np.linalg.norm(L[0]) / np.linalg.norm(L[1]))
    m.compute_correlation_scaled("samples")
    assert m.adata.obsp["X_correlation"].shape == (12, 12)
    with pytest.raises(ValueError):
        m.compute_correlation_scaled("features")


# ------------------------------------------------------------------ multimodal initialisation (initialize.py:387-470)

from salamander_amd.initialization import initialize_mmcorrnmf  # noqa: E402


@pytest.fixture
def mdata():
    rng = np.random.default_rng(1)
    names = [f"s{i}" for i in range(9)]
    a = sal.AnnData(rng.poisson(20.0, size=(9, 96)).astype(float), obs_names=names)
    b = sal.AnnData(rng.poisson(15.0, size=(9, 83)).astype(float), obs_names=names)
    return sal.MuData({"sbs": a, "indel": b})


def test_initialize_mmcorrnmf_defaults(mdata):
    np.random.seed(0)
    asigs, variance = initialize_mmcorrnmf(mdata, [2, 3], 2, "flat")
    assert list(asigs) == ["sbs", "indel"]
    assert asigs["sbs"].X.shape == (2, 96) and asigs["indel"].X.shape == (3, 83)
    assert list(asigs["indel"].obs_names) == ["indel Sig1", "indel Sig2", "indel Sig3"]
    assert mdata.obsm["embeddings"].shape == (9, 2) and variance == 1.0
    for name in ("sbs", "indel"):
        assert "embeddings" not in mdata[name].obsm  # sample embeddings are shared, not per modality
        assert np.array_equal(mdata[name].obs["scalings"].values, np.zeros(9))
        assert asigs[name].obsm["embeddings"].shape[1] == 2


def test_initialize_mmcorrnmf_given(mdata):
    U = np.ones((9, 2))
    given_sig = sal.AnnData(np.full((1, 96), 1 / 96), obs_names=["SBS1"], var_names=mdata["sbs"].var_names)
    asigs, variance = initialize_mmcorrnmf(
        mdata, [2, 2], 2, "flat", {"sbs": {"asignatures": given_sig, "signature_scalings": np.array([1.0, 2.0])}, "sample_embeddings": U, "variance": 2}
    )
    assert list(asigs["sbs"].obs_names) == ["SBS1", "sbs Sig1"]
    assert np.array_equal(asigs["sbs"].obs["scalings"].values, [1.0, 2.0])
    assert mdata.obsm["embeddings"] is U and variance == 2.0


@pytest.mark.parametrize(
    "given,exc",
    [({"rna": {}}, ValueError), ({"sbs": {"sample_embeddings": np.zeros((9, 2))}}, KeyError), ({"indel": {"variance": 1.0}}, KeyError),
     ({"sbs": {"signature_scalings": np.zeros(5)}}, ValueError)],
)
def test_initialize_mmcorrnmf_rejects(mdata, given, exc):
    with pytest.raises(exc):
        initialize_mmcorrnmf(mdata, [2, 2], 2, "flat", given)


def test_multimodal_model_setup_checks(mdata):
    from salamander_amd.models import MultimodalCorrNMF

    m = MultimodalCorrNMF(ns_signatures=[2, 3])
    assert m.dim_embeddings == 3 and m.mod_names == ["mod1", "mod2"] and m.objective == "maximize"
    with pytest.raises(TypeError):
        m._setup_mdata({"sbs": None})
    with pytest.raises(ValueError, match="modalities"):
        MultimodalCorrNMF(ns_signatures=[2])._setup_mdata(mdata)
    bad = sal.MuData({"a": mdata["sbs"], "b": sal.AnnData(np.ones((9, 4)))})
    with pytest.raises(ValueError, match="sample names"):
        m._setup_mdata(bad)
    m._setup_mdata(mdata)
    assert m.mod_names == ["sbs", "indel"] and m.sample_names == [f"s{i}" for i in range(9)]


def test_corr_models_accept_the_distributed_flag():
    from salamander_amd.models import MultimodalCorrNMF

    """Sample-sharded fitting is supported (tests/test_distributed_gloo.py runs it on two ranks): the flag is kept."""
    assert CorrNMFDet(n_signatures=2, dim_embeddings=2, distributed=True).distributed
    assert MultimodalCorrNMF([2, 3], dim_embeddings=2, distributed=True).distributed
    assert not MultimodalCorrNMF([2, 3]).distributed
