"""Initialisation of ``(signatures, exposures)`` on the MI355X engine.

The reference initialises on the host (``src/salamander/initialization/methods.py:58-135``); its default,
``init_method="nndsvd"``, is scikit-learn's private ``_initialize_nmf`` (``methods.py:83``): a *randomized* rank-K
SVD of the ``n_samples x n_features`` count matrix followed by the NNDSVD sign split.  At 10^5 - 10^6 samples that
host SVD costs seconds, far more than the device-resident fit that follows.  With at most 96 features the SVD is
cheap to do exactly:

* ``X^T X`` (96 x 96) in one pass over the resident ``X`` on the fp64 MFMA units (``salnmf_init_gram``; all-reduced
  over the shards of a distributed fit), its eigendecomposition on the host (``numpy.linalg.eigh``, < 1 ms):
  eigenvectors = right singular vectors ``V``, eigenvalues = ``sigma^2``;
* the left singular vectors ``U = X V / sigma`` as a projection on the device (``salnmf_init_project``), which also
  returns the squared norms of the positive / negative parts of every column -- all the sign split needs from ``U``;
* the sign split itself, sklearn's ``W[W < eps] = 0``, the ``nndsvda`` fill and the reference's post-processing
  (``normalize_WH`` + clip, ``initialize.py:116-118``) per element on the device (``salnmf_init_finish``); the
  signature side (K x 96) in NumPy here, statement by statement as sklearn does it.

What differs from the reference: the SVD is exact instead of randomized (``_randomized_svd`` with the *global*
NumPy RNG, so the reference's own result changes from call to call unless ``init_kwargs={"seed": ...}``).  NNDSVD
does not depend on the sign convention of the singular vectors, so wherever the randomized SVD has converged the two
agree entry by entry (the reference's 96 x 10 fixtures: to rounding); for the trailing, nearly degenerate singular
vectors of a noisy matrix they span the same subspace.  Callers that need sklearn's seeded randomized SVD pass
``init_kwargs={"seed": s}`` (or ``device_init=False``) and get the host path.  ``random``, ``nndsvdar`` and the
exposures of ``separableNMF`` draw from NumPy's legacy global RNG and stay on the host for the same reason.
"""

from __future__ import annotations

import sys

import numpy as np

from .utils import EPSILON

DEVICE_METHODS = ("flat", "nndsvd", "nndsvda")
_SKLEARN_EPS = 1e-6  # `eps` of sklearn's _initialize_nmf: smaller entries are zeroed


def _norm(v):
    return float(np.sqrt(np.dot(v, v)))


def nndsvd_signature_side(eigvals, eigvecs, pos2, neg2, n_signatures, x_mean, method):
    """The K x V half of sklearn's NNDSVD and the per-column recipe for the N x K half.

    ``eigvals``/``eigvecs``: top-K eigenpairs of ``X^T X`` (descending); ``pos2``/``neg2``: squared norms of the
    positive / negative parts of the columns of ``U = X V / sigma``.  Returns ``(S_raw (K, V), scale (K),
    take_neg (K), fill)`` with ``E_raw[:, j] = fill(threshold(scale_j * part_j(U[:, j])))``.
    """
    K = n_signatures
    S = np.sqrt(np.maximum(eigvals, 0.0))  # singular values
    Vt = eigvecs.T  # (K, V)
    H = np.zeros_like(Vt)
    scale = np.zeros(K)
    take_neg = np.zeros(K, dtype=np.int32)
    H[0, :] = np.sqrt(S[0]) * np.abs(Vt[0, :])
    scale[0] = np.sqrt(S[0])
    for j in range(1, K):
        y = Vt[j, :]
        y_p, y_n = np.maximum(y, 0), np.abs(np.minimum(y, 0))
        x_p_nrm, x_n_nrm = np.sqrt(pos2[j]), np.sqrt(neg2[j])
        y_p_nrm, y_n_nrm = _norm(y_p), _norm(y_n)
        m_p, m_n = x_p_nrm * y_p_nrm, x_n_nrm * y_n_nrm
        if m_p > m_n:
            x_nrm, v, sigma = x_p_nrm, y_p / y_p_nrm, m_p
        else:
            x_nrm, v, sigma, take_neg[j] = x_n_nrm, y_n / y_n_nrm, m_n, 1
        lbd = np.sqrt(S[j] * sigma)
        scale[j] = lbd / x_nrm
        H[j, :] = lbd * v
    H[H < _SKLEARN_EPS] = 0
    fill = 0.0
    if method == "nndsvda":
        fill = float(x_mean)
        H[H == 0] = fill
    return H, scale, take_neg, fill


_blas_controller = None


def _single_blas_thread():
    """Context that limits NumPy's BLAS / LAPACK to one thread.  The controller is built once per process: finding the
    loaded BLAS libraries walks the process's shared objects (1.5 ms, a third of a c2 initialisation)."""
    global _blas_controller
    try:
        from threadpoolctl import ThreadpoolController
    except ImportError:  # pragma: no cover - threadpoolctl comes with scikit-learn
        import contextlib

        return contextlib.nullcontext()
    # (rebuilt when the process has loaded more modules since: a BLAS that arrives later -- SciPy's OpenBLAS -- is not in a
    # controller built before it, and limit() would silently not apply to it)
    n_modules = len(sys.modules)
    if _blas_controller is None or _blas_controller[1] != n_modules:
        _blas_controller = (ThreadpoolController(), n_modules)
    return _blas_controller[0].limit(limits=1, user_api="blas")


def initialize_on_device(engine, n_signatures, method, given_signatures_mat=None, n_samples_total=None):
    """``initialize_mat`` (``initialize.py:44-119``) for the deterministic methods on the engine's resident ``X``.

    Leaves the exposures resident in the engine's ``H`` and returns ``signatures (K, V)`` (normalised, clipped); the
    caller downloads the exposures if it wants them on the host.  ``n_samples_total``: samples over all shards
    (``X.mean()`` of the ``nndsvda`` fill).
    """
    if method not in DEVICE_METHODS:
        raise ValueError(f"init method '{method}' is not available on the device (one of {DEVICE_METHODS}).")
    K, V = n_signatures, engine.V
    n_total = engine.N if n_samples_total is None else n_samples_total
    plan = None
    if method == "flat":
        S = np.full((K, V), 1.0 / V)
    else:
        G, x_sum = engine.init_gram()
        # one BLAS thread for the 96 x 96 eigenproblem: a multi-threaded LAPACK call leaves its worker pool spinning on
        # every host core for a while, which on a busy or small host starves the HIP runtime's threads -- the first
        # ~100 steps after the initialisation then stall for 30-80 ms (tools/t_bisect.py)
        with _single_blas_thread():
            evals, evecs = np.linalg.eigh(G)
        order = np.argsort(evals)[::-1][:K]
        evals, evecs = evals[order], evecs[:, order]
        sigma = np.sqrt(np.maximum(evals, 0.0))
        with np.errstate(divide="ignore", invalid="ignore"):
            B = (evecs / sigma).T  # (K, V): U = X @ B.T
        B = np.nan_to_num(B, nan=0.0, posinf=0.0, neginf=0.0)  # rank-deficient X: sklearn's U columns are noise there
        pos2, neg2 = engine.init_project(np.ascontiguousarray(B))
        with np.errstate(divide="ignore", invalid="ignore"):
            S, scale, take_neg, fill = nndsvd_signature_side(evals, evecs, pos2, neg2, K, x_sum / (n_total * V), method)
        plan = (scale, take_neg, fill)
    if given_signatures_mat is not None:
        g = given_signatures_mat.shape[0]
        S[:g, :] = given_signatures_mat.copy()
    # normalize_WH + clip (initialize.py:116-118): W / colsum, H * colsum
    with np.errstate(divide="ignore", invalid="ignore"):
        colsum = S.sum(axis=1)
        S_out = (S / colsum[:, None]).clip(EPSILON)
    if method == "flat":
        engine.init_flat(colsum)
    else:
        scale, take_neg, fill = plan
        engine.init_finish(np.nan_to_num(scale, nan=0.0, posinf=0.0), take_neg, colsum, _SKLEARN_EPS, fill)
    return S_out
