"""NumPy restatement of the dense pieces of Salamander's correlated NMF (the oracle, row f1).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  Never on the product path.

Parity status: **pinned** for everything below (``tests/test_oracle_corrnmf.py``) by
(i) the reference's own fixtures ``tests/test_data/models/{corrnmf,multimodal_corrnmf}/*`` (committed as data
under ``tests/golden/ref_fixtures/``): objective (ELBO), aux, both scalings, both embedding updates, variance,
signatures; and (ii) vectors produced by executing the reference's ``_utils_corrnmf`` functions in the build
container on larger problems (``tests/golden/make_golden.py --corr`` -> ``tests/golden/corr_synth.npz``:
K = 7 and 12, every function and a three-update CorrNMFDet trajectory).
The embedding update delegates to ``scipy.optimize.minimize(method="Newton-CG")`` exactly
as the reference does (``_utils_corrnmf.py:400-407``); SciPy is a third-party dependency of
the reference (pinned 1.13.1 in its ``poetry.lock``, 1.15.3 installed here), so beyond the
fixtures (K <= 2, N = 10) the embedding step is pinned only to the installed SciPy.

Array conventions follow the reference's call sites: ``data_mat (N, V)``,
``signatures_mat (K, V)``, ``exposures_mat (N, K)``, embeddings ``(K | N, dim)``.
Citations are ``file:line`` under ``/root/reference/src/salamander/models/``.
"""

from __future__ import annotations

import numpy as np

EPSILON = np.finfo(np.float32).eps


def compute_exposures(signature_scalings, sample_scalings, signature_embeddings, sample_embeddings) -> np.ndarray:
    """exposures[n, k] = exp(beta_k + alpha_n + <L_k, U_n>)  (``_utils_corrnmf.py:11-25``)."""
    beta = np.asarray(signature_scalings, dtype=np.float64)
    alpha = np.asarray(sample_scalings, dtype=np.float64)
    L = np.asarray(signature_embeddings, dtype=np.float64)
    U = np.asarray(sample_embeddings, dtype=np.float64)
    logits = beta[:, None] + alpha[None, :] + L @ U.T  # (K, N)
    return np.exp(logits).T


def compute_aux(data_mat, signatures_mat, exposures_mat) -> np.ndarray:
    """aux[k, n] = H[n, k] * sum_v W[k, v] * X[n, v] / (H W)[n, v]  (``_utils_corrnmf.py:28-52``)."""
    X = np.asarray(data_mat, dtype=np.float64)
    W = np.asarray(signatures_mat, dtype=np.float64)
    H = np.asarray(exposures_mat, dtype=np.float64)
    ratios = X / (H @ W)
    return H.T * (W @ ratios.T)


def poisson_llh(X, W, H) -> float:
    """Poisson log-likelihood for real-valued counts (``_utils_klnmf.py:98-160``), X (V, N), W (V, K), H (K, N).

    ``sum_{WH != 0} X log(WH) - sum WH - sum gammaln(1 + X)``.
    """
    from scipy.special import gammaln

    X = np.asarray(X, dtype=np.float64)
    WH = np.asarray(W, dtype=np.float64) @ np.asarray(H, dtype=np.float64)
    nz = WH != 0
    logs = np.where(nz, X * np.log(np.where(nz, WH, 1.0)), 0.0)
    return float(logs.sum() - WH.sum() - gammaln(1.0 + X).sum())


def elbo_corrnmf(
    data_mat, signatures_mat, exposures_mat, signature_embeddings, sample_embeddings, variance, penalize_sample_embeddings=True
) -> float:
    """Evidence lower bound (``_utils_corrnmf.py:55-100``): Poisson term + Gaussian priors of the embeddings."""
    L = np.asarray(signature_embeddings, dtype=np.float64)
    U = np.asarray(sample_embeddings, dtype=np.float64)
    K, dim = L.shape
    value = poisson_llh(np.asarray(data_mat).T, np.asarray(signatures_mat).T, np.asarray(exposures_mat).T)
    value -= 0.5 * dim * K * np.log(2 * np.pi * variance)
    value -= np.sum(L**2) / (2 * variance)
    if penalize_sample_embeddings:
        value -= 0.5 * dim * U.shape[0] * np.log(2 * np.pi * variance)
        value -= np.sum(U**2) / (2 * variance)
    return float(value)


def update_signature_scalings(aux, sample_scalings, signature_embeddings, sample_embeddings) -> np.ndarray:
    """beta_k = log sum_n aux[k, n] - log sum_n exp(alpha_n + <L_k, U_n>)  (``_utils_corrnmf.py:103-138``)."""
    aux = np.asarray(aux, dtype=np.float64)
    alpha = np.asarray(sample_scalings, dtype=np.float64)
    L = np.asarray(signature_embeddings, dtype=np.float64)
    U = np.asarray(sample_embeddings, dtype=np.float64)
    return np.log(aux.sum(axis=1)) - np.log(np.exp(alpha[None, :] + L @ U.T).sum(axis=1))


def update_sample_scalings(data_mat, signature_scalings, signature_embeddings, sample_embeddings) -> np.ndarray:
    """alpha_n = log sum_v X[n, v] - log sum_k exp(beta_k + <L_k, U_n>)  (``_utils_corrnmf.py:141-179``)."""
    X = np.asarray(data_mat, dtype=np.float64)
    beta = np.asarray(signature_scalings, dtype=np.float64)
    L = np.asarray(signature_embeddings, dtype=np.float64)
    U = np.asarray(sample_embeddings, dtype=np.float64)
    return np.log(X.sum(axis=1)) - np.log(np.exp(beta[:, None] + L @ U.T).sum(axis=0))


# --------------------------------------------------------------------------- embeddings (SciPy Newton-CG)


def embedding_objective(x, others, scaling, scalings_other, variance, aux_vector) -> float:
    """Negative surrogate objective of one embedding (``_utils_corrnmf.py:182-239``)."""
    sp = others @ x
    value = float(np.dot(sp, aux_vector))
    value -= float(np.sum(np.exp(scaling + scalings_other + sp)))
    value -= float(np.dot(x, x)) / (2 * variance)
    return -value


def embedding_gradient(x, others, scaling, scalings_other, variance, summand_grad) -> np.ndarray:
    """Negative gradient (``_utils_corrnmf.py:242-293``)."""
    weights = np.exp(scaling + scalings_other + others @ x)
    grad = -(weights[:, None] * others).sum(axis=0) + summand_grad - x / variance
    return -grad


def embedding_hessian(x, others, scaling, scalings_other, variance) -> np.ndarray:
    """Negative Hessian (``_utils_corrnmf.py:296-351``): sum_i w_i o_i o_i^T + I / variance."""
    weights = np.exp(scaling + scalings_other + others @ x)
    return (others * weights[:, None]).T @ others + np.eye(len(x)) / variance


def update_embedding(embedding_init, others, scaling, scalings_other, variance, aux_vec, **kwargs) -> np.ndarray:
    """One embedding by SciPy's Newton-CG, then push tiny entries away from zero (``_utils_corrnmf.py:354-410``)."""
    from scipy import optimize

    others = np.asarray(others, dtype=np.float64)
    aux_vec = np.asarray(aux_vec, dtype=np.float64)
    summand_grad = (aux_vec[:, None] * others).sum(axis=0)
    res = optimize.minimize(
        fun=lambda x: embedding_objective(x, others, scaling, scalings_other, variance, aux_vec),
        x0=np.array(embedding_init, dtype=np.float64),
        method="Newton-CG",
        jac=lambda x: embedding_gradient(x, others, scaling, scalings_other, variance, summand_grad),
        hess=lambda x: embedding_hessian(x, others, scaling, scalings_other, variance),
        **kwargs,
    )
    x = res.x
    x[(0 < x) & (x < EPSILON)] = EPSILON
    x[(-EPSILON < x) & (x < 0)] = -EPSILON
    return x


def update_signature_embeddings(aux, signature_embeddings, sample_embeddings, signature_scalings, sample_scalings, variance):
    """All signature embeddings, one SciPy solve each, in order (``corrnmf_det.py:88-113``)."""
    L = np.array(signature_embeddings, dtype=np.float64)
    for k in range(L.shape[0]):
        L[k] = update_embedding(L[k], sample_embeddings, signature_scalings[k], sample_scalings, variance, aux[k])
    return L


def update_sample_embeddings(aux, signature_embeddings, sample_embeddings, signature_scalings, sample_scalings, variance):
    """All sample embeddings, ``maxiter=3`` each (``corrnmf_det.py:115-141``)."""
    U = np.array(sample_embeddings, dtype=np.float64)
    for n in range(U.shape[0]):
        U[n] = update_embedding(
            U[n], signature_embeddings, sample_scalings[n], signature_scalings, variance, aux[:, n], options={"maxiter": 3}
        )
    return U


def update_variance(signature_embeddings, sample_embeddings) -> float:
    """Mean square of all embedding entries, clipped from below (``corrnmf_det.py:60-69``)."""
    both = np.concatenate([np.asarray(signature_embeddings), np.asarray(sample_embeddings)])
    return float(np.clip(np.mean(both**2), EPSILON, None))


def corrnmf_det_step(X, W, beta, alpha, L, U, variance, n_given=0):
    """One ``CorrNMFDet._update_parameters`` (``corrnmf_det.py:157-169``) on plain arrays.

    X (N, V), W (K, V); returns the updated ``(W, beta, alpha, L, U, variance, exposures)``.
    """
    from . import klnmf_oracle as kl

    alpha = update_sample_scalings(X, beta, L, U)
    H = compute_exposures(beta, alpha, L, U)
    aux = compute_aux(X, W, H)
    beta = update_signature_scalings(aux, alpha, L, U)
    L_new = update_signature_embeddings(aux, L, U, beta, alpha, variance)
    U_new = update_sample_embeddings(aux, L_new, U, beta, alpha, variance)
    variance = update_variance(L_new, U_new)
    W_new = kl.update_W(X.T, W.T, H.T, n_given_signatures=n_given).T
    return W_new, beta, alpha, L_new, U_new, variance, H


# --------------------------------------------------------------------------- multimodal CorrNMF (mmcorrnmf.py)
# Per modality: data X_m (N, V_m), signatures W_m (K_m, V_m), scalings beta_m (K_m,), alpha_m (N,),
# signature embeddings L_m (K_m, dim).  Shared: sample embeddings U (N, dim) and the variance.
# Pinned by the reference's fixtures tests/test_data/models/multimodal_corrnmf/* (tests/test_oracle_corrnmf.py).


def mm_elbo(Xs, Ws, Hs, Ls, U, variance) -> float:
    """``MultimodalCorrNMF.objective_function`` (``mmcorrnmf.py:168-194``)."""
    U = np.asarray(U, dtype=np.float64)
    value = sum(elbo_corrnmf(X, W, H, L, U, variance, penalize_sample_embeddings=False) for X, W, H, L in zip(Xs, Ws, Hs, Ls))
    value -= 0.5 * U.shape[1] * U.shape[0] * np.log(2 * np.pi * variance)
    value -= np.sum(U**2) / (2 * variance)
    return float(value)


def mm_update_sample_embeddings(auxs, Ls, U, betas, alphas, variance) -> np.ndarray:
    """Joint solve per sample over the concatenated signatures of all modalities (``mmcorrnmf.py:398-428``).

    The per-term "scaling" is the sample's scaling in the term's modality (``:412-418``)."""
    L_all = np.concatenate([np.asarray(L, dtype=np.float64) for L in Ls])
    beta_all = np.concatenate([np.asarray(b, dtype=np.float64) for b in betas])
    aux_all = np.concatenate([np.asarray(a, dtype=np.float64) for a in auxs])
    U = np.array(U, dtype=np.float64)
    for n in range(U.shape[0]):
        scalings = np.concatenate([np.repeat(alpha[n], len(b)) for alpha, b in zip(alphas, betas)])
        U[n] = update_embedding(U[n], L_all, scalings, beta_all, variance, aux_all[:, n], options={"maxiter": 3})
    return U


def mm_update_variance(Ls, U) -> float:
    """``mmcorrnmf.py:305-317``."""
    both = np.concatenate([np.concatenate([np.asarray(L) for L in Ls]), np.asarray(U)])
    return float(np.clip(np.mean(both**2), EPSILON, None))


def mm_step(Xs, Ws, betas, alphas, Ls, U, variance, n_given=None):
    """One ``MultimodalCorrNMF._update_parameters`` (``mmcorrnmf.py:443-453``) on plain arrays."""
    from . import klnmf_oracle as kl

    n_given = [0] * len(Xs) if n_given is None else n_given
    alphas = [update_sample_scalings(X, b, L, U) for X, b, L in zip(Xs, betas, Ls)]
    Hs = [compute_exposures(b, a, L, U) for b, a, L in zip(betas, alphas, Ls)]
    auxs = [compute_aux(X, W, H) for X, W, H in zip(Xs, Ws, Hs)]
    betas = [update_signature_scalings(aux, a, L, U) for aux, a, L in zip(auxs, alphas, Ls)]
    Ls_new = [update_signature_embeddings(aux, L, U, b, a, variance) for aux, L, b, a in zip(auxs, Ls, betas, alphas)]
    U_new = mm_update_sample_embeddings(auxs, Ls_new, U, betas, alphas, variance)
    variance = mm_update_variance(Ls_new, U_new)
    Ws_new = [kl.update_W(X.T, W.T, H.T, n_given_signatures=g).T for X, W, H, g in zip(Xs, Ws, Hs, n_given)]
    return Ws_new, betas, alphas, Ls_new, U_new, variance, Hs
