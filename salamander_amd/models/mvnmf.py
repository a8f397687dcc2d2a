"""``MvNMF``: minimum-volume KL NMF on the MI355X engine.

Drop-in for ``src/salamander/models/mvnmf.py`` (constructor ``:116-137``, objective
``:149-156``, ``_update_H``/``_update_W`` ``:162-195``, ``_update_parameters`` ``:197-210``,
``_setup_fitting_parameters`` ``:212-218``).  The (K x K) inverse / log-determinant, the
closed-form W root and the backtracking line search run on the device; ``_gamma`` persists
across iterations exactly as in the reference.
"""

from __future__ import annotations

from typing import Any, Literal

import numpy as np

from ._utils_klnmf import _engine_for
from .standard_nmf import StandardNMF


# -- the function-level entry points of the reference module, on the device: same names, argument order and shapes
# (X (V, N), W (V, K), H (K, N)) as src/salamander/models/mvnmf.py:19-92.  Each call uploads its arguments to a fresh engine
# and downloads the result; inputs are never modified.  The device-resident loop lives in MvNMF.fit.
def volume_logdet(W: np.ndarray, delta: float) -> float:
    """``log det(W^T W + delta I)`` (mvnmf.py:19-24)."""
    W = np.asarray(W, dtype=np.float64)
    V, K = W.shape
    e = _engine_for(np.ones((V, 1)), W, np.ones((K, 1)))
    try:
        return e.mv_logdet(delta)
    finally:
        e.close()


def kl_divergence_penalized(X: np.ndarray, W: np.ndarray, H: np.ndarray, lam: float, delta: float) -> float:
    """``KL(X || W H) + lam * volume_logdet(W, delta)`` (mvnmf.py:27-34)."""
    e = _engine_for(X, W, H)
    try:
        return e.mv_objective(lam, delta)
    finally:
        e.close()


def update_W_unconstrained(X: np.ndarray, W: np.ndarray, H: np.ndarray, lam: float, delta: float, n_given_signatures: int = 0) -> np.ndarray:
    """The closed-form root of the min-volume W step before the line search (mvnmf.py:37-66): ``(V, K)``."""
    e = _engine_for(X, W, H)
    try:
        return e.mv_update_W_unconstrained(n_given_signatures, lam, delta).T
    finally:
        e.close()


def line_search(X: np.ndarray, W: np.ndarray, H: np.ndarray, lam: float, delta: float, gamma: float, W_unconstrained: np.ndarray):
    """Backtracking between ``W`` and ``W_unconstrained`` (mvnmf.py:69-92): ``(W_new, H_new, gamma)``."""
    e = _engine_for(X, W, H)
    try:
        gamma = e.mv_line_search(lam, delta, gamma, np.asarray(W_unconstrained, dtype=np.float64).T)
        return e.download_W().T, e.download_H().T, gamma
    finally:
        e.close()


class MvNMF(StandardNMF):
    def __init__(
        self,
        n_signatures: int = 1,
        init_method: str = "nndsvd",
        lam: float = 1.0,
        delta: float = 1.0,
        min_iterations: int = 500,
        max_iterations: int = 10000,
        conv_test_freq: int = 10,
        tol: float = 1e-7,
        **engine_kwargs,
    ):
        super().__init__(
            n_signatures, init_method, min_iterations, max_iterations, conv_test_freq, tol, **engine_kwargs
        )
        self.lam = lam
        self.delta = delta
        self._gamma = 1.0

    @property
    def objective(self) -> Literal["minimize", "maximize"]:
        return "minimize"

    def compute_reconstruction_errors(self) -> None:
        self._sync_to_device()
        self.adata.obs["reconstruction_error"] = self._engine.samplewise_kl()

    def objective_function(self) -> float:
        self._sync_to_device()
        self._objective_after_steps = None  # (whatever a fit loop left behind: the state may have been edited since)
        return self._device_objective()

    # -- the reference's single-step hooks (tests/test_mvnmf.py:70-76)
    def _update_H(self) -> None:
        self._sync_to_device()
        self._engine.update_H()
        self.adata.obsm["exposures"] = self._engine.download_H()

    def _update_W_unconstrained(self, n_given_signatures: int = 0) -> np.ndarray:
        """``(V, K)``, as the reference's hook returns it (mvnmf.py:167-175)."""
        self._sync_to_device()
        return self._engine.mv_update_W_unconstrained(n_given_signatures, self.lam, self.delta).T

    def _line_search(self, W_unconstrained: np.ndarray) -> None:
        """mvnmf.py:177-188: updates the signatures, the exposures and ``_gamma``."""
        self._sync_to_device()
        self._gamma = self._engine.mv_line_search(self.lam, self.delta, self._gamma, np.asarray(W_unconstrained, dtype=np.float64).T)
        self._sync_from_device()

    def _update_W(self, n_given_signatures: int = 0) -> None:
        if n_given_signatures == self.n_signatures:
            return
        self._sync_to_device()
        self._gamma = self._engine.mv_update_W(n_given_signatures, self.lam, self.delta, self._gamma)
        self._sync_from_device()

    def _update_parameters(self, given_parameters: dict[str, Any] | None = None) -> None:
        self._sync_to_device()
        self._device_steps(1, given_parameters)
        self._sync_from_device()

    # -- device-resident pieces used by fit()
    def _device_steps(self, n_steps: int, given_parameters) -> None:
        # the last step's line search has evaluated the objective of the state the steps leave behind (its accepted value,
        # mvnmf.py:82-89): kept for the convergence test that follows in fit()
        # (inside fit the next block of steps normally follows: the engine keeps its speculative first half across the calls)
        self._gamma, self._objective_after_steps = self._engine.mv_step_objective(
            n_steps, self._n_given(given_parameters), self.lam, self.delta, self._gamma, more_follows=getattr(self, "_fit_running", False)
        )

    def _device_objective(self) -> float:
        cached, self._objective_after_steps = getattr(self, "_objective_after_steps", None), None
        if cached is not None:
            return cached
        return self._engine.mv_objective(self.lam, self.delta)

    def _setup_fitting_parameters(self, fitting_kwargs: dict[str, Any] | None = None) -> None:
        self._objective_after_steps = None
        self._gamma = 1.0
