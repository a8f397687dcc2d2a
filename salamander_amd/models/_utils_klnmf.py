"""Function-level entry points with the reference's names and shapes, on the device.

Same signatures as ``src/salamander/models/_utils_klnmf.py`` -- ``X (V, N)``,
``W (V, K)``, ``H (K, N)``, optional per-sample weights -- so the reference's unit tests
(``tests/test_utils_klnmf.py:48-196``) read the same against this module.  Each call
uploads its arguments to a fresh engine, runs the HIP kernels through the C ABI and
downloads the result; inputs are never modified.  The device-resident loop lives in
``SignatureNMF.fit``; these are the single-shot forms.
"""

from __future__ import annotations

import numpy as np

from .. import _lib
from ..engine import Engine

EPSILON = np.finfo(np.float32).eps
GIVEN_PARAMETERS_ALLOWED = ["asignatures"]


def _engine_for(X, W, H, weights_kl=None, weights_lhalf=None, device=0) -> Engine:
    X, W, H = np.asarray(X, dtype=np.float64), np.asarray(W, dtype=np.float64), np.asarray(H, dtype=np.float64)
    V, N = X.shape
    K = W.shape[1]
    if W.shape != (V, K) or H.shape != (K, N):
        raise ValueError("Incompatible shapes: X (V, N), W (V, K), H (K, N) expected.")
    e = Engine(N, V, K, device=device)
    e.upload_X(X.T)
    e.upload_W(W.T)
    e.upload_H(H.T)
    e.set_weights(weights_kl, weights_lhalf)
    return e


def kl_divergence(X, W, H, weights=None) -> float:
    """Generalised KL divergence D(X || WH) with optional per-sample weights (:11-55)."""
    e = _engine_for(X, W, H, weights, None)
    try:
        return e.objective()
    finally:
        e.close()


def samplewise_kl_divergence(X, W, H, weights=None) -> np.ndarray:
    """Per-sample KL divergence (:58-97)."""
    e = _engine_for(X, W, H)
    try:
        errors = e.samplewise_kl()
    finally:
        e.close()
    if weights is not None:
        errors = errors * weights
    return errors


def poisson_llh(X, W, H) -> float:
    """Poisson log-likelihood generalised to non-negative real counts, log-factorial terms included (:136-160)."""
    e = _engine_for(X, W, H)
    try:
        return e.corr_poisson_llh()
    finally:
        e.close()


def update_W(X, W, H, weights_kl=None, n_given_signatures: int = 0) -> np.ndarray:
    """W step, clipping only the non-given columns (:164-217)."""
    e = _engine_for(X, W, H, weights_kl, None)
    try:
        e.update_W(n_given_signatures, _lib.CLIP_NON_GIVEN)
        return e.download_W().T
    finally:
        e.close()


def update_H(X, W, H, weights_kl=None, weights_lhalf=None) -> np.ndarray:
    """H step with the current W (:220-278)."""
    e = _engine_for(X, W, H, weights_kl, weights_lhalf)
    try:
        e.update_H()
        return e.download_H().T
    finally:
        e.close()


def update_WH(X, W, H, weights_kl=None, weights_lhalf=None, n_given_signatures: int = 0):
    """The joint KLNMF step (:281-361): returns ``(W_updated, H_updated)``."""
    e = _engine_for(X, W, H, weights_kl, weights_lhalf)
    try:
        e.kl_step(1, n_given_signatures)
        return e.download_W().T, e.download_H().T
    finally:
        e.close()
