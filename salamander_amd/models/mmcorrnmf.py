"""``MultimodalCorrNMF``: several correlated NMF models over the same samples, fitted jointly on the
MI355X engine.

Drop-in for ``src/salamander/models/mmcorrnmf.py``: every modality has its own data, signatures,
scalings and signature embeddings; the sample embeddings and the variance are shared
(``mmcorrnmf.py:1-7``).  One engine per modality holds that modality's state; the shared sample
embeddings are mirrored in every engine and updated by one joint device solve per sample
(``salnmf_corr_update_sample_embeddings_multi``).  ``fit`` keeps everything resident; one update
(``_update_parameters``, ``:443-453``) is, in this order and for all modalities: sample scalings,
exposures, aux, signature scalings, signature embeddings, then the shared sample embeddings, the
variance, and the signatures.
"""

from __future__ import annotations

from typing import Any, Literal

import numpy as np
import pandas as pd

from .. import _lib
from ..anndata_compat import MUDATA_TYPES, AnnData, MuData
from ..engine import Engine
from ..device_init import DEVICE_METHODS, initialize_on_device
from ..initialization import INIT_METHODS, check_given_asignatures, initialize_mmcorrnmf, package_signatures
from ..utils import EPSILON, type_checker, value_checker
from . import _utils_corrnmf
from ._utils_klnmf import update_W


def _given(given_parameters):
    return {} if given_parameters is None else given_parameters


class MultimodalCorrNMF:
    def __init__(
        self,
        ns_signatures: list[int],
        dim_embeddings: int | None = None,
        init_method: str = "nndsvd",
        min_iterations: int = 500,
        max_iterations: int = 10000,
        conv_test_freq: int = 10,
        tol: float = 1e-7,
        *,
        device: int = 0,
        distributed: bool = False,
        device_init: bool = True,
    ):
        value_checker("init_method", init_method, INIT_METHODS)
        self.ns_signatures = ns_signatures
        self.dim_embeddings = int(np.max(ns_signatures)) if dim_embeddings is None else dim_embeddings
        self.init_method = init_method
        self.min_iterations = min_iterations
        self.max_iterations = max_iterations
        self.conv_test_freq = conv_test_freq
        self.tol = tol
        self.variance = 1.0
        # ours: which GPU, and whether mdata is this rank's shard of the samples (one communicator per modality)
        self.device = device
        self.distributed = distributed
        self.device_init = device_init  # deterministic init methods: the signatures on the GPU (device_init.py)
        # ours: the modalities' signature-embedding solves driven from one host thread each (same launches per engine, same
        # bits; `_device_steps`); False: one modality after the other (sharded models always)
        self.solve_side_by_side = True
        self._pool = None
        self._comm_attached: dict[str, bool] = {}
        self._x_resident: set[str] = set()  # modalities whose X the device already holds from the initialisation
        names = [f"mod{n}" for n in range(1, len(ns_signatures) + 1)]
        self.mdata = MuData({name: AnnData() for name in names})
        self.asignatures = {name: AnnData() for name in names}
        self.history: dict[str, Any] = {}
        self._engines: dict[str, Engine] = {}

    # ------------------------------------------------------------------ accessors (mmcorrnmf.py:69-104)
    @property
    def mod_names(self) -> list[str]:
        return list(self.mdata.mod.keys())

    @property
    def mutation_types(self) -> dict[str, list[str]]:
        return {name: list(adata.var_names) for name, adata in self.mdata.mod.items()}

    @property
    def signature_names(self) -> dict[str, list[str]]:
        return {name: list(asigs.obs_names) for name, asigs in self.asignatures.items()}

    @property
    def sample_names(self) -> list[str]:
        return list(self.mdata.obs_names)

    @property
    def signatures(self) -> dict[str, pd.DataFrame]:
        return {name: asigs.to_df() for name, asigs in self.asignatures.items()}

    @property
    def exposures(self) -> dict[str, pd.DataFrame]:
        return {
            name: pd.DataFrame(self.mdata[name].obsm["exposures"], index=self.sample_names, columns=self.asignatures[name].obs_names)
            for name in self.mod_names
        }

    @property
    def objective(self) -> Literal["minimize", "maximize"]:
        return "maximize"

    def _given_mod(self, given_parameters, mod_name) -> dict:
        return _given(given_parameters).get(mod_name, {})

    # ------------------------------------------------------------------ per-parameter methods on the AnnData state
    def compute_exposures(self) -> None:
        for name in self.mod_names:
            adata, asigs = self.mdata[name], self.asignatures[name]
            adata.obsm["exposures"] = _utils_corrnmf.compute_exposures(
                asigs.obs["scalings"].values, adata.obs["scalings"].values, asigs.obsm["embeddings"], self.mdata.obsm["embeddings"]
            )

    def compute_reconstruction(self) -> None:
        for name in self.mod_names:
            e = _utils_corrnmf._klnmf_engine(self.mdata[name].X, self.asignatures[name].X, self.mdata[name].obsm["exposures"])
            try:
                self.mdata[name].obsm["X_reconstructed"] = e.reconstruct()
            finally:
                e.close()

    @property
    def data_reconstructed(self) -> dict[str, pd.DataFrame]:
        if any("X_reconstructed" not in adata.obsm for adata in self.mdata.mod.values()):
            self.compute_reconstruction()
        return {
            name: pd.DataFrame(adata.obsm["X_reconstructed"], index=adata.obs_names, columns=adata.var_names)
            for name, adata in self.mdata.mod.items()
        }

    def compute_reconstruction_errors(self) -> None:
        self.compute_exposures()
        for name in self.mod_names:
            adata = self.mdata[name]
            e = _utils_corrnmf._klnmf_engine(adata.X, self.asignatures[name].X, adata.obsm["exposures"])
            try:
                adata.obs["reconstruction_error"] = e.samplewise_kl()
            finally:
                e.close()
        self.mdata.update()

    @property
    def reconstruction_errors(self) -> dict[str, float]:
        if any("reconstruction_error" not in self.mdata[name].obs for name in self.mod_names):
            self.compute_reconstruction_errors()
        return {name: float(np.sum(adata.obs["reconstruction_error"])) for name, adata in self.mdata.mod.items()}

    @property
    def reconstruction_error(self) -> float:
        return float(np.sum(list(self.reconstruction_errors.values())))

    def objective_function(self) -> float:
        """The ELBO: every modality's data and signature-embedding terms, the sample-embedding prior once (:168-194)."""
        U = self.mdata.obsm["embeddings"]
        elbo = 0.0
        for name in self.mod_names:
            adata, asigs = self.mdata[name], self.asignatures[name]
            elbo += _utils_corrnmf.elbo_corrnmf(
                adata.X, asigs.X, adata.obsm["exposures"], asigs.obsm["embeddings"], U, self.variance, penalize_sample_embeddings=False
            )
        elbo -= 0.5 * self.dim_embeddings * self.mdata.n_obs * np.log(2 * np.pi * self.variance)
        elbo -= np.sum(np.asarray(U) ** 2) / (2 * self.variance)
        return float(elbo)

    def _setup_mdata(self, mdata) -> None:
        type_checker("mdata", mdata, MUDATA_TYPES)
        if mdata.n_mod != len(self.ns_signatures):
            raise ValueError(f"The data has to have {len(self.ns_signatures)} many modalities.")
        expected = list(mdata.mod.values())[0].obs_names
        for adata in mdata.mod.values():
            if not all(adata.obs_names == expected):
                raise ValueError("The sample names of the different modalities are not identical.")
        self.mdata = mdata

    def _initialize(self, given_parameters: dict[str, Any] | None = None, init_kwargs: dict[str, Any] | None = None) -> None:
        init_kwargs = {} if init_kwargs is None else init_kwargs.copy()
        base = None
        if self.device_init and self.init_method in DEVICE_METHODS and "seed" not in init_kwargs:
            base = self._device_base
        self.asignatures, self.variance = initialize_mmcorrnmf(
            self.mdata, self.ns_signatures, self.dim_embeddings, self.init_method, given_parameters, base=base, **init_kwargs
        )
        self.compute_exposures()

    def _engine_for(self, name: str, N: int, V: int, n_signatures: int):
        """The engine of one modality (created on first use), its communicator attached when ``distributed``."""
        e = self._engines.get(name)
        if e is None or (e.N, e.V, e.K, e.device) != (N, V, n_signatures, self.device):
            if e is not None:
                e.close()
            e = self._engines[name] = Engine(N, V, n_signatures, device=self.device)
            self._comm_attached[name] = False
            self._x_resident.discard(name)
        if self.distributed and not self._comm_attached.get(name):
            from ..distributed import attach_communicator, attach_peer_exchange

            attach_communicator(e)
            # the small all-reduces (numerators, scalings, at small dim_embeddings the lockstep solves' evaluation records) by
            # peer stores where the GPUs of the node can map each other's memory; otherwise everything stays on RCCL
            attach_peer_exchange(e, required=False)
            self._comm_attached[name] = True
        return e

    def _device_base(self, name, adata, n_signatures, method, given_asignatures=None):
        """``initialize_base`` of one modality on its engine (X stays resident for the fit)."""
        given_mat = None
        if given_asignatures is not None:
            check_given_asignatures(given_asignatures, adata, n_signatures)
            given_mat = np.asarray(given_asignatures.X)
        X = np.ascontiguousarray(adata.X, dtype=np.float64)
        e = self._engine_for(name, X.shape[0], X.shape[1], n_signatures)
        e.upload_X(X)
        n_total = int(e.comm_info()[2]) if self.distributed else X.shape[0]
        S = initialize_on_device(e, n_signatures, method, given_mat, n_total)
        self._x_resident.add(name)
        return package_signatures(adata, S, n_signatures, given_asignatures), None

    def _compute_auxs(self) -> dict[str, np.ndarray]:
        return {
            name: _utils_corrnmf.compute_aux(self.mdata[name].X, self.asignatures[name].X, self.mdata[name].obsm["exposures"])
            for name in self.mod_names
        }

    def update_sample_scalings(self, given_parameters: dict[str, Any] | None = None) -> None:
        for name in self.mod_names:
            if "sample_scalings" not in self._given_mod(given_parameters, name):
                asigs = self.asignatures[name]
                self.mdata[name].obs["scalings"] = _utils_corrnmf.update_sample_scalings(
                    self.mdata[name].X, asigs.obs["scalings"].values, asigs.obsm["embeddings"], self.mdata.obsm["embeddings"]
                )

    def update_signature_scalings(self, auxs: dict[str, np.ndarray], given_parameters: dict[str, Any] | None = None) -> None:
        for name in self.mod_names:
            if "signature_scalings" not in self._given_mod(given_parameters, name):
                asigs = self.asignatures[name]
                asigs.obs["scalings"] = _utils_corrnmf.update_signature_scalings(
                    auxs[name], self.mdata[name].obs["scalings"].values, asigs.obsm["embeddings"], self.mdata.obsm["embeddings"]
                )

    def update_variance(self, given_parameters: dict[str, Any] | None = None) -> None:
        if "variance" not in _given(given_parameters):
            blocks = [np.asarray(asigs.obsm["embeddings"]) for asigs in self.asignatures.values()]
            blocks.append(np.asarray(self.mdata.obsm["embeddings"]))
            self.variance = np.clip(np.mean(np.concatenate(blocks) ** 2), EPSILON, None)

    def update_signatures(self, given_parameters: dict[str, Any] | None = None) -> None:
        for name in self.mod_names:
            given_mod = self._given_mod(given_parameters, name)
            n_given = given_mod["asignatures"].n_obs if "asignatures" in given_mod else 0
            adata, asigs = self.mdata[name], self.asignatures[name]
            W = update_W(
                np.asarray(adata.X).T, np.asarray(asigs.X).T, np.asarray(adata.obsm["exposures"]).T, n_given_signatures=n_given
            )
            asigs.X = W.T

    def update_signature_embeddings(self, auxs: dict[str, np.ndarray], given_parameters: dict[str, Any] | None = None) -> None:
        for name in self.mod_names:
            if "signature_embeddings" not in self._given_mod(given_parameters, name):
                asigs = self.asignatures[name]
                asigs.obsm["embeddings"] = _utils_corrnmf.update_signature_embeddings(
                    auxs[name],
                    asigs.obsm["embeddings"],
                    self.mdata.obsm["embeddings"],
                    np.asarray(asigs.obs["scalings"].values),
                    np.asarray(self.mdata[name].obs["scalings"].values),
                    self.variance,
                )

    def update_sample_embeddings(self, auxs: dict[str, np.ndarray]) -> None:
        """One joint solve per sample over the signatures of all modalities (:398-428)."""
        engines = []
        try:
            for name in self.mod_names:
                asigs = self.asignatures[name]
                engines.append(
                    _utils_corrnmf._embedding_engine(
                        auxs[name],
                        asigs.obsm["embeddings"],
                        self.mdata.obsm["embeddings"],
                        np.asarray(asigs.obs["scalings"].values),
                        np.asarray(self.mdata[name].obs["scalings"].values),
                    )
                )
            Engine.corr_update_sample_embeddings_multi(engines, self.variance, 3)
            self.mdata.obsm["embeddings"] = engines[0].corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)
        finally:
            for e in engines:
                e.close()

    def update_embeddings(self, auxs: dict[str, np.ndarray], given_parameters: dict[str, Any] | None = None) -> None:
        self.update_signature_embeddings(auxs, given_parameters)
        if "sample_embeddings" not in _given(given_parameters):
            self.update_sample_embeddings(auxs)

    def _update_parameters(self, given_parameters: dict[str, Any] | None = None) -> None:
        """One update on the AnnData / MuData state: upload, one resident step, write everything back."""
        self._sync_to_device()
        self._device_steps(1, given_parameters)
        self._sync_from_device()

    # ------------------------------------------------------------------ device-resident loop
    def _n_obs_total(self) -> int:
        """Samples over all shards when ``distributed`` (``mdata`` is this rank's shard), else ``mdata.n_obs``."""
        if self.distributed and self._engines and all(self._comm_attached.get(n) for n in self.mod_names):
            return int(self._engines[self.mod_names[0]].comm_info()[2])
        return int(self.mdata.n_obs)

    def _sync_to_device(self) -> None:
        U = np.ascontiguousarray(self.mdata.obsm["embeddings"], dtype=np.float64)
        if self.distributed:
            from ..distributed import broadcast_from_rank0

            self.variance = broadcast_from_rank0(float(self.variance))
        for name, n_signatures in zip(self.mod_names, self.ns_signatures):
            adata, asigs = self.mdata[name], self.asignatures[name]
            X = np.ascontiguousarray(adata.X, dtype=np.float64)
            N, V = X.shape
            e = self._engine_for(name, N, V, n_signatures)
            if getattr(e, "dim", None) != self.dim_embeddings:
                e.corr_configure(self.dim_embeddings)
            W = np.ascontiguousarray(asigs.X, dtype=np.float64)
            beta = np.asarray(asigs.obs["scalings"].values, dtype=np.float64)
            L = np.ascontiguousarray(asigs.obsm["embeddings"], dtype=np.float64)
            if self.distributed:
                W, beta, L = broadcast_from_rank0((W, beta, L))  # replicated parameters: rank 0's bits everywhere
            if name not in self._x_resident:
                e.upload_X(X)
            self._x_resident.discard(name)
            e.upload_W(W)
            if "exposures" in adata.obsm:
                e.upload_H(np.ascontiguousarray(adata.obsm["exposures"], dtype=np.float64))
            e.corr_upload(_lib.CORR_SIGNATURE_SCALINGS, beta)
            e.corr_upload(_lib.CORR_SAMPLE_SCALINGS, np.asarray(adata.obs["scalings"].values, dtype=np.float64))
            e.corr_upload(_lib.CORR_SIGNATURE_EMBEDDINGS, L)
            e.corr_upload(_lib.CORR_SAMPLE_EMBEDDINGS, U)

    def _device_steps(self, n_steps: int, given_parameters) -> None:
        given = _given(given_parameters)
        names = self.mod_names
        engines = [self._engines[name] for name in names]
        for _ in range(n_steps):
            for name, e in zip(names, engines):
                given_mod = given.get(name, {})
                if "sample_scalings" not in given_mod:
                    e.corr_update_sample_scalings()
                e.corr_compute_exposures()
                e.corr_compute_aux()
                if "signature_scalings" not in given_mod:
                    e.corr_update_signature_scalings()
            todo = [e for name, e in zip(names, engines) if "signature_embeddings" not in given.get(name, {})]
            if len(todo) > 1 and not self.distributed and self.solve_side_by_side:
                # the modalities' signature solves are independent (mmcorrnmf.py:319-334) and each engine has its own stream:
                # driven from one host thread per modality, one modality's evaluation rounds fill the GPU while the other's
                # small kernels between the rounds (reduction, solver replay: ~10 % of a round with the device nearly idle)
                # and its host round trips run.  Same launches per engine, hence the same bits.  (Sharded engines keep the
                # sequential order: two communicators driven concurrently would have to agree on an order across the ranks.)
                list(self._solve_pool(len(todo)).map(lambda eng: eng.corr_update_signature_embeddings(self.variance, 0), todo))
            else:
                for e in todo:
                    e.corr_update_signature_embeddings(self.variance, 0)
            if "sample_embeddings" not in given:
                Engine.corr_update_sample_embeddings_multi(engines, self.variance, 3)
            if "variance" not in given:
                self.variance = self._resident_variance()
            for name, e in zip(names, engines):
                given_mod = given.get(name, {})
                e.corr_update_signatures(given_mod["asignatures"].n_obs if "asignatures" in given_mod else 0)

    def _solve_pool(self, n: int):
        """Host threads that drive the modalities' signature solves side by side (ctypes releases the GIL inside a call)."""
        pool = self._pool
        if pool is None or pool._max_workers < n:
            from concurrent.futures import ThreadPoolExecutor

            pool = self._pool = ThreadPoolExecutor(max_workers=n, thread_name_prefix="salnmf-solve")
        return pool

    def _resident_sumsq(self):
        sig = [self._engines[name].corr_embedding_sumsq() for name in self.mod_names]
        return sum(s[0] for s in sig), sig[0][1]

    def _resident_variance(self) -> float:
        ss_sig, ss_samples = self._resident_sumsq()
        count = (sum(self.ns_signatures) + self._n_obs_total()) * self.dim_embeddings
        return float(np.clip((ss_sig + ss_samples) / count, EPSILON, None))

    def _device_objective(self) -> float:
        dim, var = self.dim_embeddings, self.variance
        log_norm = np.log(2 * np.pi * var)
        value = 0.0
        for name, n_signatures in zip(self.mod_names, self.ns_signatures):
            e = self._engines[name]
            value += e.corr_poisson_llh()
            value -= 0.5 * dim * n_signatures * log_norm + e.corr_embedding_sumsq()[0] / (2 * var)
        value -= 0.5 * dim * self._n_obs_total() * log_norm + self._resident_sumsq()[1] / (2 * var)
        return float(value)

    def _sync_from_device(self) -> None:
        for name in self.mod_names:
            e, adata, asigs = self._engines[name], self.mdata[name], self.asignatures[name]
            asigs.X = e.download_W()
            adata.obsm["exposures"] = e.download_H()
            asigs.obs["scalings"] = e.corr_download(_lib.CORR_SIGNATURE_SCALINGS)
            adata.obs["scalings"] = e.corr_download(_lib.CORR_SAMPLE_SCALINGS)
            asigs.obsm["embeddings"] = e.corr_download(_lib.CORR_SIGNATURE_EMBEDDINGS)
        self.mdata.obsm["embeddings"] = self._engines[self.mod_names[0]].corr_download(_lib.CORR_SAMPLE_EMBEDDINGS)

    # ------------------------------------------------------------------ fit (mmcorrnmf.py:455-491)
    def fit(
        self,
        mdata,
        given_parameters: dict[str, Any] | None = None,
        init_kwargs: dict[str, Any] | None = None,
        history: bool = True,
        verbose: Literal[0, 1] = 0,
        verbosity_freq: int = 100,
    ) -> "MultimodalCorrNMF":
        self._setup_mdata(mdata)
        self._initialize(given_parameters, init_kwargs)
        self._sync_to_device()
        of_values = [self._device_objective()]
        n_iteration = 0
        converged = False
        while not converged:
            n_iteration += 1
            if verbose and n_iteration % verbosity_freq == 0:
                print(f"iteration: {n_iteration}; objective: {of_values[-1]:.2f}")
            self._device_steps(1, given_parameters)
            if n_iteration % self.conv_test_freq == 0:
                prev = of_values[-1]
                of_values.append(self._device_objective())
                rel_change = np.abs(prev - of_values[-1]) / np.abs(prev)
                converged = bool(rel_change < self.tol and n_iteration >= self.min_iterations)
            converged |= n_iteration >= self.max_iterations
        self._sync_from_device()
        if history:
            self.history["objective_function"] = of_values[1:]
        self.mdata.update()
        return self

    def _out_of_scope(self, *args, **kwargs):
        raise NotImplementedError(
            "plotting / post-hoc analysis helpers are outside the scope of salamander_amd "
            "(SURVEY.md section 2); use the reference package on the fitted objects."
        )

    plot_history = plot_signatures = plot_exposures = plot_correlation = plot_embeddings = reorder = _out_of_scope
