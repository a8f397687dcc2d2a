"""Synthetic mutation-count problems of the benchmark configurations.

There is no network for datasets, so ``bench.py`` and the timing tools measure on counts drawn here:
``W_true ~ Dirichlet(0.5)`` signatures, ``H_true ~ Gamma(0.5)`` exposures scaled to ``mean_mutations`` per
sample, ``X ~ Poisson(H_true W_true)`` clipped at EPSILON as ``SignatureNMF._setup_adata`` does, and a
``random``-style initialisation ``(W0, H0)`` (``methods.py:89-109`` with a ``Generator``), normalised and clipped.
"""

from __future__ import annotations

import numpy as np

from .utils import EPSILON


def synthetic_problem(n_features: int, n_samples: int, n_signatures: int, seed: int = 0, mean_mutations: float = 2000.0):
    """Sample-major ``X (N, V)``, ``W0 (K, V)``, ``H0 (N, K)``, float64 -- AnnData's storage layout."""
    rng = np.random.default_rng(seed)
    V, N, K = n_features, n_samples, n_signatures
    signatures = rng.dirichlet(np.full(V, 0.5), size=K)
    exposures = rng.gamma(0.5, 2.0 * mean_mutations / K, size=(N, K))
    X = np.clip(rng.poisson(exposures @ signatures).astype(np.float64), EPSILON, None)
    W0 = rng.dirichlet(np.ones(V), size=K)
    H0 = X.sum(axis=1)[:, None] * rng.dirichlet(np.ones(K), size=N)
    scale = W0.sum(axis=1)
    return X, np.clip(W0 / scale[:, None], EPSILON, None), np.clip(H0 * scale[None, :], EPSILON, None)
