"""``StandardNMF``: models parameterised by a signature and an exposure matrix.

Mirrors ``src/salamander/models/standard_nmf.py:32-58`` (``_initialize``).  The embedding
plots of the reference class are out of scope.
"""

from __future__ import annotations

from typing import Any

import numpy as np

from ..device_init import DEVICE_METHODS, initialize_on_device
from ..initialization import GIVEN_PARAMETERS_STANDARD_NMF, check_given_asignatures, initialize_standard_nmf, package_signatures
from ..utils import dict_checker
from .signature_nmf import SignatureNMF


class StandardNMF(SignatureNMF):
    _background_setup = True  # _initialize below joins the background clip before host code reads adata.X

    def _initialize(self, given_parameters: dict[str, Any] | None = None, init_kwargs: dict[str, Any] | None = None) -> None:
        """Initialise signatures and exposures; given signatures are never overwritten later.

        The deterministic methods run on the GPU (``device_init.py``: exact SVD through the 96 x 96 Gram matrix
        instead of sklearn's randomized one) unless ``device_init=False`` or a ``seed`` is passed, which asks for
        the reference's seeded host computation (``initialize.py:221-255``)."""
        init_kwargs = {} if init_kwargs is None else init_kwargs.copy()
        # (more than 96 features: the Gram matrix is formed block pair by block pair, separableNMF's deflation walks the rows
        # over all feature blocks; more than 64 signatures: the projection and the post-processing chunk by chunk)
        on_device = self.device_init
        if on_device and self.init_method in DEVICE_METHODS and "seed" not in init_kwargs:
            if init_kwargs:
                raise TypeError(f"init method '{self.init_method}' takes no keyword arguments besides 'seed': {sorted(init_kwargs)}")
            self._initialize_on_device(given_parameters)
            return
        if on_device and self.init_method == "separableNMF" and not self.distributed and "chosen" not in init_kwargs:
            # the K deflation rounds of the signature selection on the device (deterministic: the same indices with or
            # without a seed); the exposures are the host's legacy-RNG draw, as in the reference (methods.py:133)
            n_obs, n_vars = np.shape(self.adata.X)
            e = self._ensure_engine(n_obs, n_vars, self.n_signatures)
            self._upload_X(e)
            chosen, norms = e.init_separable(self.n_signatures, return_norms=True)
            # once the selected samples span the data (rank-deficient or duplicated rows) the remaining norms are rounding
            # noise and the device's argmax need not be np.argmax's: such inputs keep the reference's host selection
            if np.all(np.isfinite(norms)) and np.all(norms > 1e-12 * norms[0]):
                init_kwargs["chosen"] = chosen
            self._resident = {"X"}
        if self.init_method == "custom" and getattr(self, "_defer_exposures", False) and not self.distributed and self.n_signatures <= 64:
            self._initialize_custom_on_device(given_parameters, init_kwargs)
            return
        if self.init_method != "custom":  # ("custom" looks at the shape of X only)
            self._finish_setup()  # the host methods read the clipped adata.X
        self.asignatures = initialize_standard_nmf(
            self.adata, self.n_signatures, self.init_method, given_parameters, **init_kwargs
        )

    def _initialize_custom_on_device(self, given_parameters, init_kwargs) -> None:
        """``init_method="custom"`` inside ``fit``: the checks and the signature side of ``initialize_mat`` on the host
        (K x V), the exposure side -- ``H * colsum`` and the clip, a pass over N x K -- on the device, applied lazily by
        the first update (``Engine.set_H_scale``).  Same values as ``initialize_mat`` (the same product and clip per
        entry); the normalised initial exposures are never materialised on the host: ``fit`` replaces
        ``adata.obsm["exposures"]`` with the fitted ones anyway."""
        from ..initialization import init_custom
        from ..utils import EPSILON, type_checker

        given_parameters = {} if given_parameters is None else given_parameters.copy()
        dict_checker("given_parameters", given_parameters, GIVEN_PARAMETERS_STANDARD_NMF)
        given = given_parameters.get("asignatures")
        n_obs, n_vars = np.shape(self.adata.X)
        from types import SimpleNamespace

        S, E = init_custom(SimpleNamespace(shape=(n_obs, n_vars)), self.n_signatures, **init_kwargs)  # (only the shape of X is looked at)
        S = np.array(S, dtype=np.float64)
        if given is not None:
            check_given_asignatures(given, self.adata, self.n_signatures)
            given_mat = np.asarray(given.X)
            type_checker("given_signatures_mat", given_mat, np.ndarray)
            S[: given_mat.shape[0], :] = given_mat.copy()
        colsum = S.T.sum(axis=0)
        S_out = (S.T / colsum).clip(EPSILON).T
        e = self._ensure_engine(n_obs, n_vars, self.n_signatures)
        self._upload_X(e)
        e.upload_H(np.ascontiguousarray(E, dtype=np.float64))
        e.set_H_scale(colsum)
        self.asignatures = package_signatures(self.adata, S_out, self.n_signatures, given)
        self.adata.obsm.pop("exposures", None)  # fit() downloads the fitted exposures at its end
        self._resident = {"X", "H"}

    def _initialize_on_device(self, given_parameters) -> None:
        given_parameters = {} if given_parameters is None else given_parameters.copy()
        dict_checker("given_parameters", given_parameters, GIVEN_PARAMETERS_STANDARD_NMF)
        given = given_parameters.get("asignatures")
        given_mat = None
        if given is not None:
            check_given_asignatures(given, self.adata, self.n_signatures)
            given_mat = np.asarray(given.X)
        n_obs, n_vars = np.shape(self.adata.X)
        e = self._ensure_engine(n_obs, n_vars, self.n_signatures)
        self._upload_X(e)  # (inside fit: the raw matrix, clipped on the device, while the host's clipped copy is being made)
        S = initialize_on_device(e, self.n_signatures, self.init_method, given_mat, self._n_obs_total())
        self.asignatures = package_signatures(self.adata, S, self.n_signatures, given)
        if getattr(self, "_defer_exposures", False):
            self.adata.obsm.pop("exposures", None)  # fit() downloads the fitted exposures at its end
        else:
            self.adata.obsm["exposures"] = e.download_H()
        self._resident = {"X", "H"}
