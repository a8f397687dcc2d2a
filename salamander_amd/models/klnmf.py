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

    # -- fitting kwargs (klnmf.py:108-153)
    def _check_weights(self, weights: np.ndarray, name: str = "weights") -> None:
        type_checker(name, weights, np.ndarray)
        shape_checker(name, weights, (self.adata.n_obs,))
        if not all(weights >= 0):
            raise ValueError("Only non-negative KL-divergence and sparsity penalty weights are allowed.")

    def _setup_fitting_parameters(self, fitting_kwargs: dict[str, Any] | None = None) -> None:
        if fitting_kwargs is None:
            fitting_kwargs = {name: None for name in _FITTING_KWARGS}
        for name in fitting_kwargs:
            if name not in _FITTING_KWARGS:
                raise ValueError(
                    f"The given fitting keyword arguments include parameters outside of {_FITTING_KWARGS}."
                )
        for name, weights in fitting_kwargs.items():
            if weights is not None:
                type_checker(name, weights, [float, int, list, np.ndarray])
                if type(weights) in (float, int):
                    weights = weights * np.ones(self.adata.n_obs)
                if type(weights) is list:
                    weights = np.array(weights)
                self._check_weights(weights, name)
            setattr(self, name, weights)
