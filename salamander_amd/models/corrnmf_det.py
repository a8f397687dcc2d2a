"""``CorrNMFDet``: deterministic batch correlated NMF on the MI355X engine.

Drop-in for ``src/salamander/models/corrnmf_det.py``.  One update (``_update_parameters``,
``:157-169``) is, in this order: sample scalings, exposures, aux, signature scalings, signature
embeddings, sample embeddings, variance, signatures.  Everything that is a pass over the
``n_samples x n_features`` / ``n_samples x n_signatures`` data -- both scalings, the exposures, aux,
the signature update and the Poisson term of the ELBO -- and both families of embedding solves (Newton-CG: one workgroup per
signature, one wavefront per sample; ``csrc/salnmf_newtoncg.h``) run on the device, and the whole state
stays resident there during ``fit``: per update only the two sums of squares behind the variance come
back to the host.

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

    # -- embeddings: one Newton-CG solve per row, batched on the device (corrnmf_det.py:88-141)
    def update_signature_embeddings(self, aux: np.ndarray) -> None:
        self.asignatures.obsm["embeddings"] = _utils_corrnmf.update_signature_embeddings(
            np.asarray(aux),
            self.asignatures.obsm["embeddings"],
            self.adata.obsm["embeddings"],
            np.asarray(self.asignatures.obs["scalings"].values),
            np.asarray(self.adata.obs["scalings"].values),
            self.variance,
        )

    def update_sample_embeddings(self, aux: np.ndarray) -> None:
        self.adata.obsm["embeddings"] = _utils_corrnmf.update_sample_embeddings(
            np.asarray(aux),
            self.asignatures.obsm["embeddings"],
            self.adata.obsm["embeddings"],
            np.asarray(self.asignatures.obs["scalings"].values),
            np.asarray(self.adata.obs["scalings"].values),
            self.variance,
            maxiter=3,
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
    # With ``distributed=True`` ``adata`` is this rank's shard of the samples.  The engine then all-reduces the
    # numerator of the signature update, the two sums of the signature scalings, the Poisson term and the sum of
    # squares of the sample embeddings; the signature-embedding solves (a signature embedding depends on all samples) run as
    # lockstep rounds on the local rows with every round's sums all-reduced -- from 2 048 samples per rank on; smaller
    # cohorts gather U / alpha / aux once per update and every rank solves all of them on identical inputs.
    def _resident_variance(self) -> float:
        """update_variance on the resident embeddings (corrnmf_det.py:65-69)."""
        ss_sig, ss_samples = self._engine.corr_embedding_sumsq()
        count = (self.n_signatures + self._n_obs_total()) * self.dim_embeddings
        return float(np.clip((ss_sig + ss_samples) / count, EPSILON, None))

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
            if "signature_embeddings" not in given:
                e.corr_update_signature_embeddings(self.variance, 0)
            if "sample_embeddings" not in given:
                e.corr_update_sample_embeddings(self.variance, 3)
            if "variance" not in given:
                self.variance = self._resident_variance()
            e.corr_update_signatures(self._n_given(given))

    def _device_objective(self) -> float:
        """ELBO of the resident state; the exposures are those of the last update, as in the reference's loop."""
        ss_sig, ss_samples = self._engine.corr_embedding_sumsq()
        dim, var = self.dim_embeddings, self.variance
        log_norm = np.log(2 * np.pi * var)
        value = self._engine.corr_poisson_llh()
        value -= 0.5 * dim * self.n_signatures * log_norm + ss_sig / (2 * var)
        value -= 0.5 * dim * self._n_obs_total() * log_norm + ss_samples / (2 * var)
        return float(value)

    def _sync_from_device(self) -> None:
        super()._sync_from_device()
        e = self._engine
        self.asignatures.obs["scalings"] = e.corr_download(_lib.CORR_SIGNATURE_SCALINGS)
        self.adata.obs["scalings"] = e.corr_download(_lib.CORR_SAMPLE_SCALINGS)
        self.asignatures.obsm["embeddings"] = e.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS)
        self.adata.obsm["embeddings"] = e.corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
