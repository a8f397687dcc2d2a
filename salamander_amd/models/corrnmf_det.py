"""``CorrNMFDet``: deterministic batch correlated NMF on the MI355X engine.

Drop-in for ``src/salamander/models/corrnmf_det.py``.  One update (``_update_parameters``,
``:157-169``) is, in this order: sample scalings, exposures, aux, signature scalings, signature
embeddings, sample embeddings, variance, signatures.  Everything that is a pass over the
``n_samples x n_features`` / ``n_samples x n_signatures`` data -- both scalings, the exposures, aux,
the signature update and the Poisson term of the ELBO -- and the ``n_samples`` sample-embedding solves (Newton-CG, one
wavefront per sample) run on the device and stay resident there during ``fit``.  The
``n_signatures`` signature-embedding solves are the reference's SciPy Newton-CG calls
(``_utils_corrnmf.update_embedding``) on the host: per update ``aux``, the scalings and the sample
embeddings travel to the host and the new signature embeddings back.

The public per-parameter methods (``update_sample_scalings`` ... ``update_signatures``) act on the
AnnData state one call at a time, as the reference's tests drive them (``tests/test_corrnmf.py:128-175``).
"""

from __future__ import annotations

from typing import Any

import numpy as np

from .. import _lib
from ..utils import EPSILON
from . import _utils_corrnmf
from ._utils_klnmf import update_W
from .corrnmf import CorrNMF


def _given(given_parameters):
    return {} if given_parameters is None else given_parameters


class CorrNMFDet(CorrNMF):
    # ------------------------------------------------------------------ per-parameter updates on the AnnData state
    def _compute_aux(self) -> np.ndarray:
        return _utils_corrnmf.compute_aux(self.adata.X, self.asignatures.X, self.adata.obsm["exposures"])

    def update_sample_scalings(self, given_parameters: dict[str, Any] | None = None) -> None:
        if "sample_scalings" not in _given(given_parameters):
            self.adata.obs["scalings"] = _utils_corrnmf.update_sample_scalings(
                self.adata.X,
                self.asignatures.obs["scalings"].values,
                self.asignatures.obsm["embeddings"],
                self.adata.obsm["embeddings"],
            )

    def update_signature_scalings(self, aux: np.ndarray, given_parameters: dict[str, Any] | None = None) -> None:
        if "signature_scalings" not in _given(given_parameters):
            self.asignatures.obs["scalings"] = _utils_corrnmf.update_signature_scalings(
                aux, self.adata.obs["scalings"].values, self.asignatures.obsm["embeddings"], self.adata.obsm["embeddings"]
            )

    def update_variance(self, given_parameters: dict[str, Any] | None = None) -> None:
        if "variance" not in _given(given_parameters):
            self.variance = self._variance_of(self.asignatures.obsm["embeddings"], self.adata.obsm["embeddings"])

    @staticmethod
    def _variance_of(signature_embeddings, sample_embeddings) -> float:
        """Mean square over all embedding entries, floored at EPSILON (corrnmf_det.py:65-69)."""
        stacked = np.concatenate([np.asarray(signature_embeddings), np.asarray(sample_embeddings)])
        return np.clip(np.mean(stacked**2), EPSILON, None)

    def update_signatures(self, given_parameters: dict[str, Any] | None = None) -> None:
        W = update_W(
            np.asarray(self.adata.X).T,
            np.asarray(self.asignatures.X).T,
            np.asarray(self.adata.obsm["exposures"]).T,
            n_given_signatures=self._n_given(given_parameters),
        )
        self.asignatures.X = W.T

    # -- embeddings: one SciPy Newton-CG solve per row, on the host (corrnmf_det.py:88-141)
    @staticmethod
    def _solve_signature_embeddings(aux, L, U, signature_scalings, sample_scalings, variance) -> np.ndarray:
        L = np.array(L, dtype=np.float64)
        U = np.asarray(U, dtype=np.float64)
        for k in range(L.shape[0]):
            L[k] = _utils_corrnmf.update_embedding(L[k], U, signature_scalings[k], sample_scalings, variance, aux[k])
        return L

    @staticmethod
    def _solve_sample_embeddings(aux, L, U, signature_scalings, sample_scalings, variance) -> np.ndarray:
        """``maxiter=3`` Newton-CG per sample (corrnmf_det.py:130-141), batched on the device."""
        return _utils_corrnmf.update_sample_embeddings(aux, L, U, signature_scalings, sample_scalings, variance, maxiter=3)

    def update_signature_embeddings(self, aux: np.ndarray) -> None:
        self.asignatures.obsm["embeddings"] = self._solve_signature_embeddings(
            np.asarray(aux),
            self.asignatures.obsm["embeddings"],
            self.adata.obsm["embeddings"],
            np.asarray(self.asignatures.obs["scalings"].values),
            np.asarray(self.adata.obs["scalings"].values),
            self.variance,
        )

    def update_sample_embeddings(self, aux: np.ndarray) -> None:
        self.adata.obsm["embeddings"] = self._solve_sample_embeddings(
            np.asarray(aux),
            self.asignatures.obsm["embeddings"],
            self.adata.obsm["embeddings"],
            np.asarray(self.asignatures.obs["scalings"].values),
            np.asarray(self.adata.obs["scalings"].values),
            self.variance,
        )

    def update_embeddings(self, aux: np.ndarray, given_parameters: dict[str, Any] | None = None) -> None:
        if "signature_embeddings" not in _given(given_parameters):
            self.update_signature_embeddings(aux)
        if "sample_embeddings" not in _given(given_parameters):
            self.update_sample_embeddings(aux)

    # ------------------------------------------------------------------ one full update
    def _update_parameters(self, given_parameters: dict[str, Any] | None = None) -> None:
        """One update on the AnnData state: upload, one resident step, write everything back."""
        self._sync_to_device()
        self._device_steps(1, given_parameters)
        self._sync_from_device()

    # ------------------------------------------------------------------ device-resident loop used by fit()
    def _sync_to_device(self) -> None:
        if self.distributed:
            # a signature embedding depends on all samples: its solve would need a host-side exchange per callback
            raise NotImplementedError("CorrNMFDet does not support sample-sharded (distributed=True) fitting.")
        super()._sync_to_device()
        # host copies of what the SciPy solves read and write between device passes
        self._L = np.array(self.asignatures.obsm["embeddings"], dtype=np.float64)
        self._U = np.array(self.adata.obsm["embeddings"], dtype=np.float64)

    def _device_steps(self, n_steps: int, given_parameters) -> None:
        given = _given(given_parameters)
        e = self._engine
        for _ in range(n_steps):
            if "sample_scalings" not in given:
                e.corr_update_sample_scalings()
            e.corr_compute_exposures()
            e.corr_compute_aux()
            if "signature_scalings" not in given:
                e.corr_update_signature_scalings()
            solve_L = "signature_embeddings" not in given
            solve_U = "sample_embeddings" not in given
            if solve_L or solve_U:
                if solve_L:
                    aux = e.corr_download(_lib.CORR_AUX).T
                    beta = e.corr_download(_lib.CORR_SIGNATURE_SCALINGS)
                    alpha = e.corr_download(_lib.CORR_SAMPLE_SCALINGS)
                    self._L = self._solve_signature_embeddings(aux, self._L, self._U, beta, alpha, self.variance)
                    e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, self._L)
                if solve_U:
                    # resident: aux, both scalings, the new signature embeddings and U are all on the device
                    e.corr_update_sample_embeddings(self.variance, 3)
                    self._U = e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
            if "variance" not in given:
                self.variance = self._variance_of(self._L, self._U)
            e.corr_update_signatures(self._n_given(given))

    def _device_objective(self) -> float:
        """ELBO of the resident state; the exposures are those of the last update, as in the reference's loop."""
        return self._engine.corr_poisson_llh() + _utils_corrnmf.embedding_priors(self._L, self._U, self.variance)

    def _sync_from_device(self) -> None:
        super()._sync_from_device()
        e = self._engine
        self.asignatures.obs["scalings"] = e.corr_download(_lib.CORR_SIGNATURE_SCALINGS)
        self.adata.obs["scalings"] = e.corr_download(_lib.CORR_SAMPLE_SCALINGS)
        self.asignatures.obsm["embeddings"] = self._L
        self.adata.obsm["embeddings"] = self._U
