"""``CorrNMF``: the shared shell of the correlated-NMF models.

Drop-in for ``src/salamander/models/corrnmf.py``: the exposures are not free parameters but
``exp(signature scaling + sample scaling + <signature embedding, sample embedding>)``
(``:66-77``); the objective is the ELBO (``:86-98``, maximised); initialisation per
``initialize_corrnmf`` (``:104-136``).  The dense pieces run on the device; see
``corrnmf_det.py`` for the update step.
"""

from __future__ import annotations

from typing import Any, Literal

import numpy as np

from .. import _lib
from ..device_init import DEVICE_METHODS, initialize_on_device
from ..initialization import check_given_asignatures, initialize_corrnmf, package_signatures
from ..utils import value_checker
from . import _utils_corrnmf
from .signature_nmf import SignatureNMF


class CorrNMF(SignatureNMF):
    def __init__(
        self,
        n_signatures: int = 1,
        init_method: str = "nndsvd",
        dim_embeddings: int | None = None,
        min_iterations: int = 500,
        max_iterations: int = 10000,
        conv_test_freq: int = 10,
        tol: float = 1e-7,
        **engine_kwargs,
    ):
        super().__init__(n_signatures, init_method, min_iterations, max_iterations, conv_test_freq, tol, **engine_kwargs)
        # embedding dimension = number of signatures covers independent signatures (corrnmf.py:60-61)
        self.dim_embeddings = n_signatures if dim_embeddings is None else dim_embeddings
        self.variance = 1.0

    @property
    def objective(self) -> Literal["minimize", "maximize"]:
        return "maximize"

    # ------------------------------------------------------------------ host state -> device
    def _sync_to_device(self) -> None:
        """X, signatures and (if present) exposures as for the KL models, plus scalings and embeddings."""
        if "exposures" not in self.adata.obsm:
            self.adata.obsm["exposures"] = np.ones((self.adata.n_obs, self.n_signatures))
        super()._sync_to_device()
        e = self._engine
        if getattr(e, "dim", None) != self.dim_embeddings:
            e.corr_configure(self.dim_embeddings)
        buffers = (_lib.CORR_SIGNATURE_SCALINGS, _lib.CORR_SAMPLE_SCALINGS, _lib.CORR_SIGNATURE_EMBEDDINGS, _lib.CORR_SAMPLE_EMBEDDINGS)
        factors = [np.asarray(values, dtype=np.float64) for values in self._factors()]
        if self.distributed:
            # the signature-side parameters and the variance are replicated: every rank starts from rank 0's bits
            from ..distributed import broadcast_from_rank0

            factors[0], factors[2], variance = broadcast_from_rank0((factors[0], factors[2], float(self.variance)))
            self.variance = variance
        for which, values in zip(buffers, factors):
            e.corr_upload(which, values)

    # ------------------------------------------------------------------ reference hooks
    def _factors(self):
        """(signature scalings, sample scalings, signature embeddings, sample embeddings) of the host state."""
        sig, smp = self.asignatures, self.adata
        return sig.obs["scalings"].values, smp.obs["scalings"].values, sig.obsm["embeddings"], smp.obsm["embeddings"]

    def compute_exposures(self) -> None:
        """``adata.obsm['exposures']`` from the scalings and embeddings (corrnmf.py:66-77)."""
        self.adata.obsm["exposures"] = _utils_corrnmf.compute_exposures(*self._factors())

    def compute_reconstruction_errors(self) -> None:
        """Samplewise KL divergences of the recomputed exposures (corrnmf.py:79-84)."""
        self.compute_exposures()
        self._sync_to_device()
        self.adata.obs["reconstruction_error"] = self._engine.samplewise_kl()

    def objective_function(self, penalize_sample_embeddings: bool = True) -> float:
        """The evidence lower bound, with the exposures as currently stored (corrnmf.py:86-98)."""
        _, _, sig_embeddings, sample_embeddings = self._factors()
        llh = _utils_corrnmf.poisson_llh(self.adata.X, self.asignatures.X, self.adata.obsm["exposures"])
        return llh + _utils_corrnmf.embedding_priors(sig_embeddings, sample_embeddings, self.variance, penalize_sample_embeddings)

    def _initialize(self, given_parameters: dict[str, Any] | None = None, init_kwargs: dict[str, Any] | None = None) -> None:
        """Signatures, scalings, embeddings and variance; then the exposures (corrnmf.py:104-136)."""
        init_kwargs = {} if init_kwargs is None else init_kwargs.copy()
        base = None
        if self.device_init and self.init_method in DEVICE_METHODS and "seed" not in init_kwargs:
            base = self._device_base  # the signatures of the deterministic methods on the GPU (device_init.py)
        self.asignatures, self.variance = initialize_corrnmf(
            self.adata, self.n_signatures, self.dim_embeddings, self.init_method, given_parameters, base=base, **init_kwargs
        )
        self.compute_exposures()

    def _device_base(self, adata, n_signatures, method, given_asignatures=None):
        """``initialize_base`` on the engine: X goes up once (and stays for the fit), the signatures come back."""
        given_mat = None
        if given_asignatures is not None:
            check_given_asignatures(given_asignatures, adata, n_signatures)
            given_mat = np.asarray(given_asignatures.X)
        X = np.ascontiguousarray(adata.X, dtype=np.float64)
        e = self._ensure_engine(X.shape[0], X.shape[1], n_signatures)
        e.upload_X(X)
        S = initialize_on_device(e, n_signatures, method, given_mat, self._n_obs_total())
        self._resident = {"X"}
        return package_signatures(adata, S, n_signatures, given_asignatures), None

    def _setup_fitting_parameters(self, fitting_kwargs: dict[str, Any] | None = None) -> None:
        """No fitting parameters (corrnmf.py:138-144)."""
        return

    def compute_correlation_scaled(self, data: Literal["samples", "signatures"] = "signatures") -> None:
        """Cosine similarities of the embeddings -> ``obsp`` (corrnmf.py:146-178)."""
        value_checker("data", data, ["samples", "signatures"])
        if "embeddings" not in self.adata.obsm:
            raise AssertionError("Computing the sample or signature correlation requires fitting the CorrNMF model.")
        holder = self.adata if data == "samples" else self.asignatures
        vectors = np.asarray(holder.obsm["embeddings"])
        unit = vectors / np.sqrt(np.sum(vectors**2, axis=1))[:, None]
        correlation = unit @ unit.T
        np.fill_diagonal(correlation, 1.0)
        if data == "samples":
            self.adata.obsp["X_correlation"] = correlation
        else:
            self.asignatures.obsp["correlation"] = correlation
