"""``KLNMF``: weighted generalised-KL NMF with normalised signatures, on the MI355X engine.

Drop-in for ``src/salamander/models/klnmf.py`` (constructor ``:34-52``, fitting kwargs
``:128-153``, objective ``:64-80``, update ``:86-106``, reconstruction errors ``:54-62``).
All arithmetic runs in ``libsalnmf.so`` (``include/salnmf.h``).
"""

from __future__ import annotations

from typing import Any, Literal

import numpy as np

from ..utils import shape_checker, type_checker
from .standard_nmf import StandardNMF

_FITTING_KWARGS = ["weights_kl", "weights_lhalf"]


class KLNMF(StandardNMF):
    def __init__(
        self,
        n_signatures: int = 1,
        init_method: str = "nndsvd",
        min_iterations: int = 500,
        max_iterations: int = 10000,
        conv_test_freq: int = 10,
        tol: float = 1e-7,
        **engine_kwargs,
    ):
        super().__init__(
            n_signatures, init_method, min_iterations, max_iterations, conv_test_freq, tol, **engine_kwargs
        )
        self.weights_kl = None
        self.weights_lhalf = None

    @property
    def objective(self) -> Literal["minimize", "maximize"]:
        return "minimize"

    # -- single-call hooks (host state -> device -> host state)
    def compute_reconstruction_errors(self) -> None:
        """Unweighted samplewise KL divergences -> ``adata.obs['reconstruction_error']``."""
        self._sync_to_device()
        self.adata.obs["reconstruction_error"] = self._engine.samplewise_kl()

    def objective_function(self) -> float:
        """Weighted KL divergence plus the l-half sparsity penalty."""
        self._sync_to_device()
        return self._device_objective()

    def _update_parameters(self, given_parameters: dict[str, Any] | None = None) -> None:
        self._sync_to_device()
        self._device_steps(1, given_parameters)
        self._sync_from_device()

    # -- device-resident pieces used by fit()
    def _device_weights(self):
        return self.weights_kl, self.weights_lhalf

    def _device_steps(self, n_steps: int, given_parameters) -> None:
        self._engine.kl_step(n_steps, self._n_given(given_parameters))

    def _device_objective(self) -> float:
        return self._engine.objective()

    def _device_can_queue(self) -> bool:
        return True

    def _device_objective_async(self, slot: int) -> bool:
        self._engine.objective_async(slot)
        return True

    def _device_objectives_read(self, first: int, count: int):
        return self._engine.objective_read(first, count)

    def _device_steps_keep(self, n_steps: int, given_parameters) -> bool:
        self._engine.kl_step_keep(n_steps, self._n_given(given_parameters))
        return True

    def _device_objective_and_steps(self, slot: int, n_steps: int, given_parameters, keep: bool) -> bool:
        if not self.objective_in_step:
            return super()._device_objective_and_steps(slot, n_steps, given_parameters, keep)
        self._engine.kl_step_objective(slot, n_steps, self._n_given(given_parameters), keep)
        return True

    def _device_rollback(self) -> None:
        self._engine.kl_rollback()

    # -- fitting kwargs: per-sample loss weights and l-half penalty weights
    def _weight_vector(self, name: str, value):
        """Normalise one fitting kwarg to ``None`` or a non-negative ``ndarray (n_obs,)``.

        Same acceptance rules and exception types as the reference
        (``klnmf.py:108-126,142-151``): scalars broadcast, lists convert, anything else that is
        not an ndarray is a ``TypeError``; wrong shape or a negative entry is a ``ValueError``.
        """
        if value is None:
            return None
        type_checker(name, value, [float, int, list, np.ndarray])
        n_obs = self.adata.n_obs
        if isinstance(value, (float, int)):
            value = np.full(n_obs, float(value))
        elif isinstance(value, list):
            value = np.array(value)
        self._check_weights(value, name)
        return value

    def _check_weights(self, weights: np.ndarray, name: str = "weights") -> None:
        type_checker(name, weights, np.ndarray)
        shape_checker(name, weights, (self.adata.n_obs,))
        if not np.all(weights >= 0):  # also rejects NaN, as the reference's `all(weights >= 0)` does
            raise ValueError("Only non-negative KL-divergence and sparsity penalty weights are allowed.")

    def _setup_fitting_parameters(self, fitting_kwargs: dict[str, Any] | None = None) -> None:
        """``fitting_kwargs`` may hold ``weights_kl`` and/or ``weights_lhalf`` (``klnmf.py:128-153``)."""
        requested = dict.fromkeys(_FITTING_KWARGS) if fitting_kwargs is None else fitting_kwargs
        unknown = [name for name in requested if name not in _FITTING_KWARGS]
        if unknown:
            raise ValueError(f"The given fitting keyword arguments include parameters outside of {_FITTING_KWARGS}.")
        for name, value in requested.items():
            setattr(self, name, self._weight_vector(name, value))
