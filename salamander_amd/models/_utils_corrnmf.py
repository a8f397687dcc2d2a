"""Function-level entry points of correlated NMF with the reference's names and shapes.

Same signatures as ``src/salamander/models/_utils_corrnmf.py``: ``data_mat (N, V)``,
``signatures_mat (K, V)``, ``exposures_mat (N, K)``, ``aux (K, N)``, embeddings
``(K | N, dim)``.  The dense pieces (exposures, aux, both scaling updates, the Poisson term of
the ELBO) run on the device through the C ABI (``salnmf_corr_*``); each call uploads its
arguments to a fresh engine and downloads the result, the resident loop lives in
``CorrNMFDet.fit``.

The reference optimises every embedding with ``scipy.optimize.minimize(method="Newton-CG")``
(``_utils_corrnmf.py:400-407``), i.e. the arithmetic of that step *is* SciPy's.  Here the solves run
on the device as well (``csrc/salnmf_newtoncg.h`` restates SciPy's truncated Newton iteration and
its two line searches): ``update_sample_embeddings`` -- one wavefront per sample -- and
``update_signature_embeddings`` -- one workgroup per signature, every evaluation a pass over all
samples.  ``update_embedding`` solves a single embedding through the same kernels.  SciPy is not
imported anywhere in this package.
"""

from __future__ import annotations

import numpy as np

from .. import _lib
from ..engine import Engine

EPSILON = np.finfo(np.float32).eps


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _corr_engine(n_samples, n_features, signature_embeddings, sample_embeddings, device=0) -> Engine:
    L, U = _f64(signature_embeddings), _f64(sample_embeddings)
    if L.ndim != 2 or U.ndim != 2 or L.shape[1] != U.shape[1] or U.shape[0] != n_samples:
        raise ValueError("Incompatible shapes: embeddings (n_signatures, dim) and (n_samples, dim) expected.")
    e = Engine(n_samples, n_features, L.shape[0], device=device)
    e.corr_configure(L.shape[1])
    e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L)
    e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U)
    return e


def compute_exposures(signature_scalings, sample_scalings, signature_embeddings, sample_embeddings) -> np.ndarray:
    """The exposure matrix ``(n_samples, n_signatures)`` (:11-25)."""
    e = _corr_engine(len(sample_scalings), 1, signature_embeddings, sample_embeddings)
    try:
        e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, _f64(signature_scalings))
        e.corr_upload(_lib.CORR_SAMPLE_SCALINGS, _f64(sample_scalings))
        e.corr_compute_exposures()
        return e.download_H()
    finally:
        e.close()


def _klnmf_engine(data_mat, signatures_mat, exposures_mat) -> Engine:
    X, W, H = _f64(data_mat), _f64(signatures_mat), _f64(exposures_mat)
    N, V = X.shape
    K = W.shape[0]
    if W.shape != (K, V) or H.shape != (N, K):
        raise ValueError("Incompatible shapes: data (N, V), signatures (K, V), exposures (N, K) expected.")
    e = Engine(N, V, K)
    e.upload_X(X)
    e.upload_W(W)
    e.upload_H(H)
    return e


def compute_aux(data_mat, signatures_mat, exposures_mat) -> np.ndarray:
    r"""``aux[k, d] = \sum_v x_vd p_vkd`` of shape ``(n_signatures, n_samples)`` (:28-52)."""
    e = _klnmf_engine(data_mat, signatures_mat, exposures_mat)
    try:
        e.corr_configure(1)
        e.corr_compute_aux()
        return e.corr_download(_lib.CORR_AUX).T
    finally:
        e.close()


def poisson_llh(data_mat, signatures_mat, exposures_mat) -> float:
    """Poisson log-likelihood of real-valued counts (``_utils_klnmf.py:136-160``), sample-major arguments."""
    e = _klnmf_engine(data_mat, signatures_mat, exposures_mat)
    try:
        return e.corr_poisson_llh()
    finally:
        e.close()


def embedding_priors(signature_embeddings, sample_embeddings, variance, penalize_sample_embeddings=True) -> float:
    """The Gaussian log-prior part of the ELBO (:93-98)."""
    L, U = np.asarray(signature_embeddings), np.asarray(sample_embeddings)
    K, dim = L.shape
    value = -0.5 * dim * K * np.log(2 * np.pi * variance) - np.sum(L**2) / (2 * variance)
    if penalize_sample_embeddings:
        value += -0.5 * dim * U.shape[0] * np.log(2 * np.pi * variance) - np.sum(U**2) / (2 * variance)
    return float(value)


def elbo_corrnmf(
    data_mat, signatures_mat, exposures_mat, signature_embeddings, sample_embeddings, variance, penalize_sample_embeddings=True
) -> float:
    """Evidence lower bound of correlated NMF (:55-100)."""
    llh = poisson_llh(data_mat, signatures_mat, exposures_mat)
    return llh + embedding_priors(signature_embeddings, sample_embeddings, variance, penalize_sample_embeddings)


def update_signature_scalings(aux, sample_scalings, signature_embeddings, sample_embeddings) -> np.ndarray:
    """New signature scalings ``(n_signatures,)`` from ``aux (n_signatures, n_samples)`` (:103-138)."""
    aux = _f64(aux)
    e = _corr_engine(aux.shape[1], 1, signature_embeddings, sample_embeddings)
    try:
        e.corr_upload(_lib.CORR_SAMPLE_SCALINGS, _f64(sample_scalings))
        e.corr_upload(_lib.CORR_AUX, _f64(aux.T))
        e.corr_update_signature_scalings()
        return e.corr_download(_lib.CORR_SIGNATURE_SCALINGS)
    finally:
        e.close()


def update_sample_scalings(data_mat, signature_scalings, signature_embeddings, sample_embeddings) -> np.ndarray:
    """New sample scalings ``(n_samples,)``; ``data_mat (n_samples, n_features)`` as at the call sites (:141-179)."""
    X = _f64(data_mat)
    e = _corr_engine(X.shape[0], X.shape[1], signature_embeddings, sample_embeddings)
    try:
        e.upload_X(X)
        e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, _f64(signature_scalings))
        e.corr_update_sample_scalings()
        return e.corr_download(_lib.CORR_SAMPLE_SCALINGS)
    finally:
        e.close()


# ----------------------------------------------------------------------------- embeddings (device Newton-CG)


def _embedding_engine(aux, signature_embeddings, sample_embeddings, signature_scalings, sample_scalings) -> Engine:
    aux = _f64(aux)
    e = _corr_engine(aux.shape[1], 1, signature_embeddings, sample_embeddings)
    try:
        e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, _f64(signature_scalings))
        e.corr_upload(_lib.CORR_SAMPLE_SCALINGS, _f64(sample_scalings))
        e.corr_upload(_lib.CORR_AUX, _f64(aux.T))
    except Exception:
        e.close()
        raise
    return e


def update_sample_embeddings(
    aux, signature_embeddings, sample_embeddings, signature_scalings, sample_scalings, variance, maxiter: int = 3
) -> np.ndarray:
    """All sample embeddings, one Newton-CG solve each with ``maxiter=3`` (``corrnmf_det.py:115-141``); ``aux (K, N)``."""
    e = _embedding_engine(aux, signature_embeddings, sample_embeddings, signature_scalings, sample_scalings)
    try:
        e.corr_update_sample_embeddings(variance, maxiter)
        return e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
    finally:
        e.close()


def update_signature_embeddings(
    aux, signature_embeddings, sample_embeddings, signature_scalings, sample_scalings, variance, maxiter: int = 0
) -> np.ndarray:
    """All signature embeddings, one Newton-CG solve each, SciPy's default iteration limit (``corrnmf_det.py:88-113``)."""
    e = _embedding_engine(aux, signature_embeddings, sample_embeddings, signature_scalings, sample_scalings)
    try:
        e.corr_update_signature_embeddings(variance, maxiter)
        return e.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS)
    finally:
        e.close()


def update_embedding(
    embedding_init, embeddings_other, scaling, scalings_other, variance, aux_vec, outer_prods_embeddings_other=None, **kwargs
) -> np.ndarray:
    """Optimise one signature or sample embedding (:354-410).

    ``embeddings_other (n_other, dim)``, ``scalings_other (n_other,)``, ``aux_vec (n_other,)``; the only
    keyword the reference passes on is ``options={"maxiter": 3}`` for sample embeddings.  ``scaling`` may be
    an array over the other embeddings (multimodal use, :208-210): it is folded into ``scalings_other``.
    """
    options = dict(kwargs.pop("options", None) or {})
    maxiter = int(options.pop("maxiter", 0) or 0)
    if kwargs or options:
        raise TypeError(f"unsupported Newton-CG arguments: {sorted(kwargs) + sorted(options)}")
    x0 = _f64(np.atleast_1d(embedding_init))[None, :]
    others = _f64(embeddings_other)
    aux_vec = _f64(np.atleast_1d(aux_vec))
    scalings_other = _f64(np.atleast_1d(scalings_other))
    if np.ndim(scaling) > 0:  # exp(scaling_i + scalings_other_i + ...): only the sum matters
        scalings_other = scalings_other + _f64(scaling)
        scaling = 0.0
    one = np.array([float(scaling)])
    if others.shape[0] <= 64:
        # few terms: the "sample" layout -- the embedding is a sample, the others are the signatures
        return update_sample_embeddings(aux_vec[:, None], others, x0, scalings_other, one, variance, maxiter)[0]
    # many terms: the "signature" layout -- the embedding is the one signature, the others are the samples
    return update_signature_embeddings(aux_vec[None, :], x0, others, one, scalings_other, variance, maxiter)[0]
