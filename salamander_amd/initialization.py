"""Initialisation of ``(W, H)`` for the standard NMF models -- the step before the hot path.

A one-time host computation; it mirrors the reference's
behaviour so that ``fit(adata, init_kwargs=...)`` is a drop-in:
``src/salamander/initialization/methods.py:15-135`` (the seven methods) and
``initialize.py:44-119,122-255`` (post-processing, given signatures, AnnData packaging).
Parity runs inject identical ``W0, H0`` through ``init_method="custom"``.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .anndata_compat import ANNDATA_TYPES, AnnData, concat_rows
from .utils import EPSILON, dict_checker, normalize_WH, shape_checker, type_checker, value_checker

INIT_METHODS = ("custom", "flat", "nndsvd", "nndsvda", "nndsvdar", "random", "separableNMF")
GIVEN_PARAMETERS_STANDARD_NMF = ["asignatures"]


def init_custom(data_mat, n_signatures, signatures_mat, exposures_mat):
    """User-supplied matrices, type- and shape-checked (methods.py:27-55)."""
    n_samples, n_features = data_mat.shape
    type_checker("signatures_mat", signatures_mat, np.ndarray)
    type_checker("exposures_mat", exposures_mat, np.ndarray)
    shape_checker("signatures_mat", signatures_mat, (n_signatures, n_features))
    shape_checker("exposures_mat", exposures_mat, (n_samples, n_signatures))
    return signatures_mat, exposures_mat


def init_flat(data_mat, n_signatures):
    """Uniform signatures; each sample's mass split evenly (methods.py:58-66)."""
    n_samples, n_features = data_mat.shape
    S = np.full((n_signatures, n_features), 1.0 / n_features)
    E = np.repeat((data_mat.sum(axis=1) / n_signatures)[:, None], n_signatures, axis=1)
    return S, E


def init_random(data_mat, n_signatures, seed=None):
    """Dirichlet signatures and scaled Dirichlet exposures from the global RNG (methods.py:89-109)."""
    if seed is not None:
        np.random.seed(seed)
    n_samples, n_features = data_mat.shape
    S = np.random.dirichlet(np.ones(n_features), size=n_signatures)
    E = data_mat.sum(axis=1)[:, None] * np.random.dirichlet(np.ones(n_signatures), size=n_samples)
    return S, E


def init_nndsvd(data_mat, n_signatures, method="nndsvd", seed=None):
    """scikit-learn's NNDSVD family, as the reference wraps it (methods.py:69-86)."""
    try:
        from sklearn.decomposition import _nmf as sknmf
    except Exception as exc:  # pragma: no cover
        raise ImportError("init_method='nndsvd*' needs scikit-learn.") from exc
    if seed is not None:
        np.random.seed(seed)
    E, S = sknmf._initialize_nmf(data_mat, n_signatures, init=method)  # pylint: disable=protected-access
    return S, E


def init_separableNMF(data_mat, n_signatures, seed=None, chosen=None):
    """Successive projection (Gillis & Vavasis 2013) for signatures + random exposures (methods.py:112-135).

    ``chosen``: the selected sample indices when the selection already ran on the device
    (``Engine.init_separable``: the same K rounds as one pass over the resident matrix each); the deflation below, K
    products with a V x N temporary, is then skipped."""
    if chosen is None:
        R = data_mat.T / data_mat.T.sum(axis=0)
        chosen = np.empty(n_signatures, dtype=int)
        for k in range(n_signatures):
            norms = (R**2).sum(axis=0)
            j = int(np.argmax(norms))
            u = R[:, j]
            R = R - np.outer(u, u @ R) / norms[j]
            chosen[k] = j
    chosen = np.asarray(chosen, dtype=int)
    S = data_mat[chosen, :].astype(float)
    S /= S.sum(axis=1, keepdims=True)
    _, E = init_random(data_mat, n_signatures, seed=seed)
    return S, E


def initialize_mat(data_mat, n_signatures, method="nndsvd", given_signatures_mat=None, **kwargs):
    """``(signatures (K, V), exposures (N, K))``: raw init, given rows, normalise, clip (initialize.py:44-119)."""
    value_checker("method", method, INIT_METHODS)
    if method == "custom":
        S, E = init_custom(data_mat, n_signatures, **kwargs)
    elif method == "flat":
        S, E = init_flat(data_mat, n_signatures)
    elif method in ("nndsvd", "nndsvda", "nndsvdar"):
        S, E = init_nndsvd(data_mat, n_signatures, method=method, **kwargs)
    elif method == "random":
        S, E = init_random(data_mat, n_signatures, **kwargs)
    else:
        S, E = init_separableNMF(data_mat, n_signatures, **kwargs)

    if given_signatures_mat is not None:
        type_checker("given_signatures_mat", given_signatures_mat, np.ndarray)
        g, vg = given_signatures_mat.shape
        if vg != data_mat.shape[1]:
            raise ValueError("The given signature matrix has a different number of features than the data.")
        if g > n_signatures:
            raise ValueError("The given signature matrix contains too many signatures.")
        S[:g, :] = given_signatures_mat.copy()

    # normalize_WH + clip (initialize.py:116-118; utils.py:155-158), with the exposures kept in their (N, K) storage
    # layout: the same products entry by entry (H[k, n] * colsum[k]), but one pass over a C-contiguous array that the
    # upload can take as it is, instead of a (K, N) temporary, its clipped copy and a transposing copy at c2's 40 MB
    colsum = S.T.sum(axis=0)
    W = (S.T / colsum).clip(EPSILON)
    E = E * colsum[None, :]
    np.clip(E, EPSILON, None, out=E)
    return W.T, E


def check_given_asignatures(given_asignatures, adata, n_signatures) -> None:
    """initialize.py:122-155."""
    type_checker("given_asignatures", given_asignatures, ANNDATA_TYPES)
    if given_asignatures.n_vars != adata.n_vars:
        raise ValueError("The given signatures have a different number of features than the data.")
    if not all(given_asignatures.var_names == adata.var_names):
        raise ValueError("The features of the given signatures and the data are not identical.")
    if given_asignatures.n_obs > n_signatures:
        raise ValueError("The number of given signatures exceeds the number of signatures to initialize.")


def initialize_base(adata, n_signatures, method="nndsvd", given_asignatures=None, **kwargs):
    """``(asignatures, exposures_mat)`` without touching ``adata`` (initialize.py:158-218)."""
    given_mat = None
    if given_asignatures is not None:
        check_given_asignatures(given_asignatures, adata, n_signatures)
        given_mat = np.asarray(given_asignatures.X)

    S, E = initialize_mat(np.asarray(adata.X), n_signatures, method, given_mat, **kwargs)
    return package_signatures(adata, S, n_signatures, given_asignatures), E


def package_signatures(adata, S, n_signatures, given_asignatures=None):
    """The signature matrix as an AnnData with the reference's names and annotations (initialize.py:205-217)."""
    asignatures = AnnData(S)
    asignatures.var_names = adata.var_names
    names = [f"Sig{k + 1}" for k in range(n_signatures)]
    asignatures.obs_names = names
    if given_asignatures is not None:
        # given signatures keep their own annotations; the rest are Sig1..Sig(K-g) (initialize.py:211-216)
        g = given_asignatures.n_obs
        asignatures.obs_names = list(np.roll(names, g))
        asignatures = concat_rows(given_asignatures, asignatures[g:, :])
    return asignatures


def initialize_standard_nmf(adata, n_signatures, method="nndsvd", given_parameters: dict[str, Any] | None = None, **kwargs):
    """Builds ``asignatures`` and writes ``adata.obsm['exposures']`` (initialize.py:221-255)."""
    given_parameters = {} if given_parameters is None else given_parameters.copy()
    dict_checker("given_parameters", given_parameters, GIVEN_PARAMETERS_STANDARD_NMF)
    asignatures, E = initialize_base(adata, n_signatures, method, given_parameters.get("asignatures"), **kwargs)
    adata.obsm["exposures"] = E
    return asignatures


# ----------------------------------------------------------------------------- correlated NMF

GIVEN_PARAMETERS_CORRNMF = [
    "asignatures",
    "signature_scalings",
    "sample_scalings",
    "signature_embeddings",
    "sample_embeddings",
    "variance",
]


def check_given_parameters_corrnmf(adata, n_signatures, dim_embeddings, given_parameters) -> None:
    """Keys, types and shapes of a priori known CorrNMF parameters (initialize.py:258-316)."""
    dict_checker("given_parameters", given_parameters, GIVEN_PARAMETERS_CORRNMF)
    if "asignatures" in given_parameters:
        check_given_asignatures(given_parameters["asignatures"], adata, n_signatures)
    expected = {
        "signature_scalings": (n_signatures,),
        "sample_scalings": (adata.n_obs,),
        "signature_embeddings": (n_signatures, dim_embeddings),
        "sample_embeddings": (adata.n_obs, dim_embeddings),
    }
    for key, shape in expected.items():
        if key in given_parameters:
            type_checker(f"given_{key}", given_parameters[key], np.ndarray)
            shape_checker(f"given_{key}", given_parameters[key], shape)
    if "variance" in given_parameters:
        type_checker("given_variance", given_parameters["variance"], [float, int])
        if given_parameters["variance"] <= 0.0:
            raise ValueError("The variance has to be a positive real number.")


def initialize_corrnmf(
    adata,
    n_signatures,
    dim_embeddings,
    method="nndsvd",
    given_parameters: dict[str, Any] | None = None,
    initialize_sample_embeddings: bool = True,
    base=None,
    **kwargs,
):
    """Signatures as for standard NMF; scalings zero; embeddings standard normal from the global RNG;
    variance 1 -- each unless given (initialize.py:319-384).  Returns ``(asignatures, variance)``.

    ``base`` (ours): a replacement for :func:`initialize_base` with the same signature -- the models pass the
    device-side initialisation of the signatures (``device_init.py``) here."""
    if method == "custom":
        raise ValueError("Custom parameter initializations are currently not supported for (multimodal) correlated NMF.")
    given_parameters = {} if given_parameters is None else given_parameters.copy()
    check_given_parameters_corrnmf(adata, n_signatures, dim_embeddings, given_parameters)
    asignatures, _ = (base or initialize_base)(adata, n_signatures, method, given_parameters.get("asignatures"), **kwargs)

    def standard_normal(n):
        return np.random.multivariate_normal(np.zeros(dim_embeddings), np.identity(dim_embeddings), size=n)

    asignatures.obs["scalings"] = given_parameters.get("signature_scalings", np.zeros(n_signatures))
    adata.obs["scalings"] = given_parameters.get("sample_scalings", np.zeros(adata.n_obs))
    if "signature_embeddings" in given_parameters:
        asignatures.obsm["embeddings"] = given_parameters["signature_embeddings"]
    else:
        asignatures.obsm["embeddings"] = standard_normal(n_signatures)
    if initialize_sample_embeddings:
        if "sample_embeddings" in given_parameters:
            adata.obsm["embeddings"] = given_parameters["sample_embeddings"]
        else:
            adata.obsm["embeddings"] = standard_normal(adata.n_obs)
    variance = float(given_parameters.get("variance", 1.0))
    return asignatures, variance


def check_given_parameters_mmcorrnmf(mdata, ns_signatures, dim_embeddings, given_parameters) -> None:
    """Per-modality dictionaries plus the shared 'sample_embeddings' / 'variance' (initialize.py:387-416)."""
    dict_checker("given_parameters", given_parameters, list(mdata.mod.keys()) + ["sample_embeddings", "variance"])
    for (mod_name, adata), n_signatures in zip(mdata.mod.items(), ns_signatures):
        given_mod = given_parameters.get(mod_name, {})
        check_given_parameters_corrnmf(adata, n_signatures, dim_embeddings, given_mod)
        if "sample_embeddings" in given_mod:
            raise KeyError(
                "The sample embeddings are shared across modalities in multimodal correlated NMF. "
                "They cannot be provided as given parameters on the modality level."
            )
        if "variance" in given_mod:
            raise KeyError(
                "The variance parameters of multimodal correlated NMF is shared across modalies. "
                "It cannot be provided as a given parameter on the modality level."
            )


def initialize_mmcorrnmf(mdata, ns_signatures, dim_embeddings, method="nndsvd", given_parameters: dict[str, Any] | None = None, base=None, **kwargs):
    """One ``initialize_corrnmf`` per modality (without sample embeddings), new signature names prefixed with the
    modality, shared sample embeddings and variance (initialize.py:419-470).  Returns ``(asignatures dict, variance)``."""
    given_parameters = {} if given_parameters is None else given_parameters.copy()
    check_given_parameters_mmcorrnmf(mdata, ns_signatures, dim_embeddings, given_parameters)
    asignatures = {}
    for (mod_name, adata), n_signatures in zip(mdata.mod.items(), ns_signatures):
        given_mod = given_parameters.get(mod_name, {})
        mod_base = (lambda *a, _name=mod_name, **kw: base(_name, *a, **kw)) if base is not None else None
        asigs, _ = initialize_corrnmf(
            adata, n_signatures, dim_embeddings, method, given_mod, initialize_sample_embeddings=False, base=mod_base, **kwargs
        )
        n_given = given_mod["asignatures"].n_obs if "asignatures" in given_mod else 0
        names = list(asigs.obs_names)
        asigs.obs_names = names[:n_given] + [f"{mod_name} {name}" for name in names[n_given:]]
        asignatures[mod_name] = asigs
    if "sample_embeddings" in given_parameters:
        mdata.obsm["embeddings"] = given_parameters["sample_embeddings"]
    else:
        mdata.obsm["embeddings"] = np.random.multivariate_normal(
            np.zeros(dim_embeddings), np.identity(dim_embeddings), size=mdata.n_obs
        )
    variance = float(given_parameters.get("variance", 1.0))
    return asignatures, variance
