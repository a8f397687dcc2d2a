"""The reference's own hot-path tests, re-run against the HIP engine on an MI355X.

Same fixtures, same assertions (``np.allclose`` defaults) as
``tests/test_utils_klnmf.py:48-196``, ``tests/test_klnmf.py:58-91`` and
``tests/test_mvnmf.py:57-90`` of the reference; only the import changes
(``salamander_amd`` instead of ``salamander``).  The pickled joint-update fixtures of
``test_klnmf.py`` are replaced by the equivalent ``.npy`` pair of ``utils_klnmf``/oracle.
"""

import os

import numpy as np
import pytest

import salamander_amd as sal
from conftest import REF_FIX, read_counts
from oracle import klnmf_oracle as orc
from salamander_amd.models import _utils_klnmf

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 2])
def n_signatures(request):
    return request.param


# ------------------------------------------------------------------ tests/test_utils_klnmf.py
D_UTILS = os.path.join(REF_FIX, "utils_klnmf")


@pytest.fixture
def counts():
    return read_counts(f"{D_UTILS}/counts.csv")


@pytest.fixture
def matrices_input(counts, n_signatures):
    return (counts.values, np.load(f"{D_UTILS}/W_nsigs{n_signatures}.npy"), np.load(f"{D_UTILS}/H_nsigs{n_signatures}.npy"))


@pytest.fixture
def weights_kl(counts):
    return 2 * np.ones(counts.shape[1])


@pytest.fixture
def weights_l_half(counts):
    return 2 * np.zeros(counts.shape[1])


def _load(name, k):
    return np.load(f"{D_UTILS}/{name}_nsigs{k}.npy")


def test_kl_divergence(matrices_input, n_signatures, weights_kl):
    want = _load("kl_divergence", n_signatures)
    assert np.allclose(_utils_klnmf.kl_divergence(*matrices_input), want)
    assert np.allclose(_utils_klnmf.kl_divergence(*matrices_input, weights_kl), 2 * want)


def test_samplewise_kl_divergence(matrices_input, n_signatures, weights_kl):
    want = _load("samplewise_kl_divergence", n_signatures)
    assert np.allclose(_utils_klnmf.samplewise_kl_divergence(*matrices_input), want)
    weights_kl[0] = 3
    got = _utils_klnmf.samplewise_kl_divergence(*matrices_input, weights_kl)
    assert np.allclose(got[0], 3 * want[0]) and np.allclose(got[1:], 2 * want[1:])


def test_poisson_llh(matrices_input, n_signatures):
    assert np.allclose(_utils_klnmf.poisson_llh(*matrices_input), _load("poisson_llh", n_signatures))


def test_update_W(matrices_input, n_signatures, weights_kl):
    want = _load("W_updated_standard", n_signatures)
    assert np.allclose(_utils_klnmf.update_W(*matrices_input), want)
    # constant loss function weights do not change the updated signatures
    assert np.allclose(_utils_klnmf.update_W(*matrices_input, weights_kl), want)


def test_given_signatures_update_W(matrices_input):
    X, W, H = matrices_input
    for g in range(1, W.shape[1] + 1):
        W_updated = _utils_klnmf.update_W(X, W.copy(), H, n_given_signatures=g)
        assert np.array_equal(W_updated[:, :g], W[:, :g])


def test_update_H(matrices_input, n_signatures, weights_kl, weights_l_half):
    want = _load("H_updated_standard", n_signatures)
    H_before = matrices_input[2].copy()
    assert np.allclose(_utils_klnmf.update_H(*matrices_input), want)
    assert np.array_equal(matrices_input[2], H_before)  # inputs are never modified
    # no l_half penalty -> identical exposure updates, independent of the loss function weights
    assert np.allclose(_utils_klnmf.update_H(*matrices_input, weights_kl, weights_l_half), want)


def test_update_WH(matrices_input, n_signatures, weights_kl, weights_l_half):
    Ww, Hw = _load("W_updated_joint", n_signatures), _load("H_updated_joint", n_signatures)
    for extra in ((), (weights_kl,), (weights_kl, weights_l_half)):
        Wg, Hg = _utils_klnmf.update_WH(*matrices_input, *extra)
        assert np.allclose(Wg, Ww) and np.allclose(Hg, Hw)


def test_given_signatures_update_WH(matrices_input):
    X, W, H = matrices_input
    for g in range(1, W.shape[1] + 1):
        W_updated, _ = _utils_klnmf.update_WH(X, W.copy(), H, n_given_signatures=g)
        assert np.array_equal(W_updated[:, :g], W[:, :g])


# ------------------------------------------------------------------ tests/test_klnmf.py
D_KL = os.path.join(REF_FIX, "klnmf")


def _adata(path):
    return sal.AnnData(read_counts(path).T)


@pytest.fixture
def kl_model(n_signatures):
    adata = _adata(f"{D_KL}/counts.csv")
    asig = sal.AnnData(np.load(f"{D_KL}/W_init_nsigs{n_signatures}.npy").T)
    asig.var_names = adata.var_names
    model = sal.models.KLNMF(n_signatures=n_signatures)
    model.adata = adata
    model.asignatures = asig
    model.adata.obsm["exposures"] = np.load(f"{D_KL}/H_init_nsigs{n_signatures}.npy").T
    return model


def test_klnmf_objective_function(kl_model, n_signatures):
    assert np.allclose(kl_model.objective_function(), np.load(f"{D_KL}/objective_init_nsigs{n_signatures}.npy"))


def test_klnmf_update_parameters(kl_model, n_signatures):
    X = read_counts(f"{D_KL}/counts.csv").values.astype(float)
    W0, H0 = kl_model.asignatures.X.T.copy(), kl_model.adata.obsm["exposures"].T.copy()
    kl_model._update_parameters()
    W1, H1 = orc.update_WH(X, W0, H0)  # what the reference's pickle holds (oracle pinned on the .npy pair)
    assert np.allclose(kl_model.asignatures.X, W1.T)
    assert np.allclose(kl_model.adata.obsm["exposures"], H1.T)


def test_klnmf_given_signatures(n_signatures):
    adata = _adata(f"{D_KL}/counts.csv")
    for g in range(1, n_signatures + 1):
        given = adata[:g, :].copy()
        given.X = given.X / np.sum(given.X, axis=1, keepdims=True)
        model = sal.models.KLNMF(n_signatures=n_signatures, min_iterations=3, max_iterations=3)
        model.fit(adata, given_parameters={"asignatures": given})
        assert np.allclose(given.X, model.asignatures.X[:g, :])


# ------------------------------------------------------------------ tests/test_mvnmf.py
D_MV = os.path.join(REF_FIX, "mvnmf")


@pytest.fixture
def mv_model(n_signatures):
    adata = _adata(f"{D_MV}/counts.csv")
    asig = sal.AnnData(np.load(f"{D_MV}/W_init_nsigs{n_signatures}.npy").T)
    asig.var_names = adata.var_names
    model = sal.models.MvNMF(n_signatures=n_signatures)
    model.adata = adata
    model.asignatures = asig
    model.adata.obsm["exposures"] = np.load(f"{D_MV}/H_init_nsigs{n_signatures}.npy").T
    model._gamma = 1.0
    return model


def test_mvnmf_objective_function(mv_model, n_signatures):
    assert np.allclose(mv_model.objective_function(), np.load(f"{D_MV}/objective_init_nsigs{n_signatures}.npy"))


def test_mvnmf_update_W(mv_model, n_signatures):
    mv_model._update_W()
    assert np.allclose(mv_model.asignatures.X, np.load(f"{D_MV}/W_updated_nsigs{n_signatures}.npy").T)
    assert mv_model._gamma == 1.0


def test_mvnmf_update_H(mv_model, n_signatures):
    mv_model._update_H()
    assert np.allclose(mv_model.adata.obsm["exposures"], np.load(f"{D_MV}/H_updated_nsigs{n_signatures}.npy").T)


def test_mvnmf_given_signatures(n_signatures):
    adata = _adata(f"{D_MV}/counts.csv")
    for g in range(1, n_signatures + 1):
        given = adata[:g, :].copy()
        given.X = given.X / np.sum(given.X, axis=1, keepdims=True)
        model = sal.models.MvNMF(n_signatures=n_signatures, min_iterations=3, max_iterations=3)
        model.fit(adata, given_parameters={"asignatures": given})
        assert np.allclose(given.X, model.asignatures.X[:g, :])
