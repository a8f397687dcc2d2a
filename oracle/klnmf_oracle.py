"""NumPy restatement of Salamander's KL-NMF / MvNMF update arithmetic (the oracle).

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.  This file is a CPU
restatement of the reference *algorithm*; it is never on the product path.

Parity status: **pinned**.  ``tests/test_oracle_golden.py`` checks every function
below against (i) the reference's own fixtures for this path
(``tests/test_data/models/{utils_klnmf,klnmf,mvnmf}/*.npy``, committed as data under
``tests/golden/ref_fixtures/``) and (ii) vectors produced by executing the
reference's functions in the build container (``tests/golden/make_golden.py``).

All functions use the reference's call-signature shapes
``X (V, N)``, ``W (V, K)``, ``H (K, N)`` (features x samples), float64, and never
modify their inputs.  Citations are ``file:line`` under ``/root/reference/``.
"""

from __future__ import annotations

import numpy as np

# src/salamander/models/_utils_klnmf.py:7 -- a float32 epsilon used as clip floor on float64 data
EPSILON = np.finfo(np.float32).eps


# --------------------------------------------------------------------------- objectives


def kl_divergence(X, W, H, weights=None) -> float:
    """Generalised KL divergence D(X || WH), optionally with per-sample weights.

    Follows ``_utils_klnmf.py:11-55``: per sample, entries with ``X == 0`` contribute
    only ``WH``; all others ``X*log(X/WH) - X + WH``.  The reference accumulates
    serially under ``fastmath``; the rounding order is unspecified there, so this
    vectorised form is compared with ``allclose`` only.
    """
    X = np.asarray(X, dtype=np.float64)
    WH = np.asarray(W, dtype=np.float64) @ np.asarray(H, dtype=np.float64)
    nz = X != 0
    safe_X = np.where(nz, X, 1.0)
    terms = np.where(nz, X * np.log(safe_X / WH) - X, 0.0) + WH
    per_sample = terms.sum(axis=0)
    if weights is not None:
        per_sample = per_sample * np.asarray(weights, dtype=np.float64)
    return float(per_sample.sum())


def samplewise_kl_divergence(X, W, H, weights=None) -> np.ndarray:
    """Per-sample KL divergence, ``_utils_klnmf.py:58-97``.

    Zeros of X are replaced by EPSILON in *both* X and WH before the log term
    (``:82-86``); the linear terms use the untouched X and ``H.T @ colsum(W)``
    (``:88-90``).
    """
    X = np.asarray(X, dtype=np.float64)
    W = np.asarray(W, dtype=np.float64)
    H = np.asarray(H, dtype=np.float64)
    zero = X == 0
    Xe = np.where(zero, EPSILON, X)
    WHe = np.where(zero, EPSILON, W @ H)
    log_term = (Xe * np.log(Xe / WHe)).sum(axis=0)
    errors = log_term - X.sum(axis=0) + H.T @ W.sum(axis=0)
    if weights is not None:
        errors = errors * np.asarray(weights, dtype=np.float64)
    return errors


def klnmf_objective(X, W, H, weights_kl=None, weights_lhalf=None) -> float:
    """``KLNMF.objective_function`` (``klnmf.py:64-80``): weighted KL + l-half penalty."""
    value = kl_divergence(X, W, H, weights_kl)
    if weights_lhalf is not None:
        value += float(np.dot(weights_lhalf, np.sqrt(H).sum(axis=0)))
    return value


# --------------------------------------------------------------------------- KL updates


def _ratio(X, W, H):
    """``aux = X / (W @ H)`` -- ``_utils_klnmf.py:206,256,328``."""
    return X / (W @ H)


def _h_from_factor(H, factor, weights_kl, weights_lhalf):
    """Shared tail of ``update_H`` (``:258-278``) and ``update_WH`` (``:343-361``).

    ``factor = W.T @ aux``.  Without an l-half penalty the update is the plain
    multiplicative one; with it, the closed-form root of the penalised problem.
    """
    if weights_lhalf is None:
        return np.clip(H * factor, EPSILON, None)
    inter = 4.0 * H * factor
    if weights_kl is not None:
        inter = inter * weights_kl**2
    disc = 0.25 * weights_lhalf**2 + inter
    Hn = 0.25 * (weights_lhalf / 2 - np.sqrt(disc)) ** 2
    if weights_kl is not None:
        Hn = Hn / weights_kl**2
    return np.clip(Hn, EPSILON, None)


def update_W(X, W, H, weights_kl=None, n_given_signatures=0) -> np.ndarray:
    """``_utils_klnmf.py:164-217``: W step; only the non-given columns are clipped (``:215``)."""
    X, W, H = (np.asarray(a, dtype=np.float64) for a in (X, W, H))
    K = W.shape[1]
    g = int(n_given_signatures)
    if g == K:
        return W.copy()
    aux = _ratio(X, W, H)
    if weights_kl is not None:
        aux = aux * weights_kl
    Wn = W * (aux @ H.T)
    Wn = Wn / Wn.sum(axis=0)
    Wn[:, :g] = W[:, :g]
    Wn[:, g:] = np.clip(Wn[:, g:], EPSILON, None)
    return Wn


def update_H(X, W, H, weights_kl=None, weights_lhalf=None) -> np.ndarray:
    """``_utils_klnmf.py:220-278``: H step with the *current* W."""
    X, W, H = (np.asarray(a, dtype=np.float64) for a in (X, W, H))
    aux = _ratio(X, W, H)
    return _h_from_factor(H, W.T @ aux, weights_kl, weights_lhalf)


def update_WH(X, W, H, weights_kl=None, weights_lhalf=None, n_given_signatures=0):
    """``_utils_klnmf.py:281-361``: the joint KLNMF step.

    * W uses the (optionally sample-weighted) ratio, is column-normalised, has its
      given columns restored, and is then clipped in *all* columns (``:338-341``).
    * H uses the **old** W and the unweighted ratio (``:343-361``).
    * If every signature is given, W is returned untouched -- not even clipped (``:330-331``).
    """
    X, W, H = (np.asarray(a, dtype=np.float64) for a in (X, W, H))
    K = W.shape[1]
    g = int(n_given_signatures)
    aux = _ratio(X, W, H)
    if g == K:
        Wn = W.copy()
    else:
        scaled = aux if weights_kl is None else weights_kl * aux
        Wn = W * (scaled @ H.T)
        Wn = Wn / Wn.sum(axis=0)
        Wn[:, :g] = W[:, :g]
        Wn = np.clip(Wn, EPSILON, None)
    Hn = _h_from_factor(H, W.T @ aux, weights_kl, weights_lhalf)
    return Wn, Hn


def normalize_WH(W, H):
    """``utils.py:155-158``: unit column sums for W, compensated in H."""
    s = np.sum(W, axis=0)
    return W / s, H * s[:, None]


# --------------------------------------------------------------------------- MvNMF


def volume_logdet(W, delta) -> float:
    """``mvnmf.py:19-24``: ``log det(W.T W + delta I)``."""
    K = W.shape[1]
    return float(np.log(np.linalg.det(W.T @ W + delta * np.eye(K))))


def kl_divergence_penalized(X, W, H, lam, delta) -> float:
    """``mvnmf.py:27-34``."""
    return kl_divergence(X, W, H) + lam * volume_logdet(W, delta)


def update_W_unconstrained(X, W, H, lam, delta, n_given_signatures=0) -> np.ndarray:
    """``mvnmf.py:37-66``: closed-form root of the min-volume majoriser, per entry."""
    X, W, H = (np.asarray(a, dtype=np.float64) for a in (X, W, H))
    K = W.shape[1]
    g = int(n_given_signatures)
    Y = np.linalg.inv(W.T @ W + delta * np.eye(K))
    WYm = W @ np.maximum(0.0, -Y)
    WYa = W @ np.abs(Y)
    r = H.sum(axis=1)
    G = _ratio(X, W, H) @ H.T
    b = r - 4.0 * lam * WYm
    root = np.sqrt(b**2 + 8.0 * lam * WYa * G)
    Wu = W * (root - b) / (4.0 * lam * WYa)
    Wu[:, :g] = W[:, :g]
    Wu[:, g:] = np.clip(Wu[:, g:], EPSILON, None)
    return Wu


def line_search(X, W, H, lam, delta, gamma, W_unconstrained):
    """``mvnmf.py:69-92``: backtracking on the penalised objective; gamma persists."""

    def trial(Wt):
        Wn, Hn = normalize_WH(Wt, H)
        Wn, Hn = np.clip(Wn, EPSILON, None), np.clip(Hn, EPSILON, None)
        return Wn, Hn, kl_divergence_penalized(X, Wn, Hn, lam, delta)

    f_prev = kl_divergence_penalized(X, W, H, lam, delta)
    Wn, Hn, f = trial(W_unconstrained)
    n_trials = 0
    while f > f_prev and gamma > 1e-16:
        gamma *= 0.8
        Wn, Hn, f = trial((1 - gamma) * W + gamma * W_unconstrained)
        n_trials += 1
    gamma = min(1.0, 1.2 * gamma)
    return Wn, Hn, gamma


def mvnmf_step(X, W, H, lam, delta, gamma, n_given_signatures=0):
    """One ``MvNMF._update_parameters`` (``mvnmf.py:197-210``): H first, then W + line search."""
    K = W.shape[1]
    H = update_H(X, W, H)
    if n_given_signatures == K:
        return W.copy(), H, gamma
    Wu = update_W_unconstrained(X, W, H, lam, delta, n_given_signatures)
    return line_search(X, W, H, lam, delta, gamma, Wu)


# --------------------------------------------------------------------------- fit loops


def _run_fit(step, objective, min_iterations, max_iterations, conv_test_freq, tol):
    """The convergence loop of ``SignatureNMF.fit`` (``signature_nmf.py:358-385``)."""
    of_values = [objective()]
    it = 0
    converged = False
    while not converged:
        it += 1
        step()
        if it % conv_test_freq == 0:
            prev = of_values[-1]
            of_values.append(objective())
            rel = abs(prev - of_values[-1]) / abs(prev)
            converged = rel < tol and it >= min_iterations
        converged = converged or it >= max_iterations
    return it, of_values[1:]


def fit_klnmf(
    X,
    W0,
    H0,
    weights_kl=None,
    weights_lhalf=None,
    n_given_signatures=0,
    min_iterations=500,
    max_iterations=10000,
    conv_test_freq=10,
    tol=1e-7,
):
    """KLNMF.fit from a given init: returns ``(W, H, n_iterations, history)``.

    ``X`` is clipped at EPSILON first, as ``_setup_adata`` does (``signature_nmf.py:281``).
    """
    state = {"W": np.array(W0, dtype=np.float64), "H": np.array(H0, dtype=np.float64)}
    X = np.clip(np.asarray(X, dtype=np.float64), EPSILON, None)

    def step():
        state["W"], state["H"] = update_WH(
            X, state["W"], state["H"], weights_kl, weights_lhalf, n_given_signatures
        )

    def objective():
        return klnmf_objective(X, state["W"], state["H"], weights_kl, weights_lhalf)

    it, hist = _run_fit(step, objective, min_iterations, max_iterations, conv_test_freq, tol)
    return state["W"], state["H"], it, hist


def fit_mvnmf(
    X,
    W0,
    H0,
    lam=1.0,
    delta=1.0,
    n_given_signatures=0,
    min_iterations=500,
    max_iterations=10000,
    conv_test_freq=10,
    tol=1e-7,
):
    """MvNMF.fit from a given init: returns ``(W, H, gamma, n_iterations, history)``."""
    state = {"W": np.array(W0, dtype=np.float64), "H": np.array(H0, dtype=np.float64), "g": 1.0}
    X = np.clip(np.asarray(X, dtype=np.float64), EPSILON, None)

    def step():
        state["W"], state["H"], state["g"] = mvnmf_step(
            X, state["W"], state["H"], lam, delta, state["g"], n_given_signatures
        )

    def objective():
        return kl_divergence_penalized(X, state["W"], state["H"], lam, delta)

    it, hist = _run_fit(step, objective, min_iterations, max_iterations, conv_test_freq, tol)
    return state["W"], state["H"], state["g"], it, hist


# --------------------------------------------------------------------------- sharded emulation


def update_WH_sharded(X, W, H, n_shards, weights_kl=None, weights_lhalf=None, n_given_signatures=0):
    """The joint step computed shard by shard over the sample axis.

    Each shard contributes a ``(V, K)`` partial of ``scaled_aux @ H.T``; the partials are
    summed (the all-reduce) and every shard then applies the identical W tail.  Used to
    check that the multi-GPU decomposition is the same algorithm.
    """
    X, W, H = (np.asarray(a, dtype=np.float64) for a in (X, W, H))
    K = W.shape[1]
    g = int(n_given_signatures)
    bounds = np.linspace(0, X.shape[1], n_shards + 1).astype(int)
    G = np.zeros_like(W)
    H_parts = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        wk = None if weights_kl is None else weights_kl[a:b]
        wl = None if weights_lhalf is None else weights_lhalf[a:b]
        aux = _ratio(X[:, a:b], W, H[:, a:b])
        G += (aux if wk is None else wk * aux) @ H[:, a:b].T
        H_parts.append(_h_from_factor(H[:, a:b], W.T @ aux, wk, wl))
    if g == K:
        Wn = W.copy()
    else:
        Wn = W * G
        Wn = Wn / Wn.sum(axis=0)
        Wn[:, :g] = W[:, :g]
        Wn = np.clip(Wn, EPSILON, None)
    return Wn, np.concatenate(H_parts, axis=1)


# --------------------------------------------------------------------------- synthetic workload


def synthetic_problem(V, N, K, seed=0, mean_mutations=2000.0):
    """The synthetic count matrix and init of SURVEY.md section 8(d) / BASELINE.md section 3.

    Returns sample-major arrays ``X (N, V)``, ``W0 (K, V)``, ``H0 (N, K)`` -- AnnData's
    storage layout, whose ``.T`` views are what the reference's functions receive.
    """
    rng = np.random.default_rng(seed)
    W_true = rng.dirichlet(np.full(V, 0.5), size=K)  # (K, V)
    H_true = rng.gamma(0.5, 2.0 * mean_mutations / K, size=(N, K))
    X = rng.poisson(H_true @ W_true).astype(np.float64)
    X = np.clip(X, EPSILON, None)
    W0 = rng.dirichlet(np.ones(V), size=K)  # (K, V) rows on the simplex
    H0 = X.sum(axis=1)[:, None] * rng.dirichlet(np.ones(K), size=N)
    s = W0.sum(axis=1)
    W0 = np.clip(W0 / s[:, None], EPSILON, None)
    H0 = np.clip(H0 * s[None, :], EPSILON, None)
    return X, W0, H0
