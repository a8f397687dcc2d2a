"""``SignatureNMF``: the model shell of the KL-NMF fit path, device resident.

Same constructor, hooks, ``fit`` signature, convergence rule and ``history`` cadence as
the reference's abstract base (``src/salamander/models/signature_nmf.py:138-146,
237-310, 315-385``).  The difference is *where the loop runs*: ``fit`` uploads
``X, W, H`` once, runs ``conv_test_freq`` update steps per host round trip on the
MI355X engine (only the objective scalar comes back) and writes ``W, H`` into the
AnnData objects once at the end.  The single-step hooks ``_update_parameters()`` and
``objective_function()`` still work on hand-populated models (they sync host state to
the device, do exactly one step, and sync back), as the reference's tests drive them
(``tests/test_klnmf.py:44-75``).

Out of scope here: plotting wrappers, ``reorder``, correlation and
dimensionality-reduction helpers.
"""

from __future__ import annotations

import threading
from abc import ABC, abstractmethod
from typing import Any, Literal

import numpy as np
import pandas as pd

from ..anndata_compat import ANNDATA_TYPES, AnnData
from ..engine import Engine
from ..initialization import INIT_METHODS
from ..utils import EPSILON, type_checker, value_checker


class SignatureNMF(ABC):
    # fit() may clip the caller's X in a worker thread while the device starts from the raw matrix (models whose
    # _initialize calls _finish_setup() before any host code reads adata.X)
    _background_setup = False

    def __init__(
        self,
        n_signatures: int = 1,
        init_method: str = "nndsvd",
        min_iterations: int = 500,
        max_iterations: int = 10000,
        conv_test_freq: int = 10,
        tol: float = 1e-7,
        *,
        device: int = 0,
        distributed: bool = False,
        device_init: bool = True,
        precision: str = "f64",
        objective_in_step: bool = True,
    ):
        value_checker("init_method", init_method, INIT_METHODS)
        self.n_signatures = n_signatures
        self.init_method = init_method
        self.min_iterations = min_iterations
        self.max_iterations = max_iterations
        self.conv_test_freq = conv_test_freq
        self.tol = tol
        # ours: which GPU, and whether adata is this rank's shard of the sample axis
        self.device = device
        self.distributed = distributed
        # ours: run the deterministic initialisation methods (flat, nndsvd, nndsvda) on the GPU (device_init.py)
        self.device_init = device_init
        # ours: "f32" = the opt-in fast mode of the KL step (Engine.set_precision; unweighted KLNMF fits only); the
        # default is the reference's fp64 arithmetic
        value_checker("precision", precision, ("f64", "f32"))
        self.precision = precision
        # ours: inside fit() the objective of a convergence test is evaluated in the launch of the update that follows it
        # (models that can: KLNMF) -- the value of objective_function() to rounding.  False: always as a pass of its own,
        # bit for bit objective_function() (and 3 % more wall clock at c2)
        self.objective_in_step = objective_in_step
        self._resident: set[str] = set()  # what the device already holds from the initialisation: "X", "H"

        self.adata = AnnData()
        self.asignatures = AnnData()
        self.history: dict[str, Any] = {}
        self._engine: Engine | None = None
        self._comm_attached = False

    # ------------------------------------------------------------------ accessors (signature_nmf.py:187-253)
    @property
    def mutation_types(self) -> list[str]:
        return list(self.adata.var_names)

    @property
    def signature_names(self) -> list[str]:
        return list(self.asignatures.obs_names)

    @property
    def sample_names(self) -> list[str]:
        return list(self.adata.obs_names)

    @property
    def signatures(self) -> pd.DataFrame:
        return self.asignatures.to_df()

    @property
    def exposures(self) -> pd.DataFrame:
        assert "exposures" in self.adata.obsm, "Learning the sample exposures requires fitting the NMF model."
        return pd.DataFrame(self.adata.obsm["exposures"], index=self.sample_names, columns=self.signature_names)

    def compute_reconstruction(self) -> None:
        """``exposures @ signatures`` on the device (signature_nmf.py:221-224)."""
        self._sync_to_device()
        self.adata.obsm["X_reconstructed"] = self._engine.reconstruct()

    @property
    def data_reconstructed(self) -> pd.DataFrame:
        if "X_reconstructed" not in self.adata.obsm:
            self.compute_reconstruction()
        return pd.DataFrame(self.adata.obsm["X_reconstructed"], index=self.sample_names, columns=self.mutation_types)

    @abstractmethod
    def compute_reconstruction_errors(self) -> None:
        """Adds samplewise errors as ``adata.obs['reconstruction_error']``."""

    @property
    def reconstruction_error(self) -> float:
        if "reconstruction_error" not in self.adata.obs:
            self.compute_reconstruction_errors()
        return float(np.sum(self.adata.obs["reconstruction_error"]))

    @property
    @abstractmethod
    def objective(self) -> Literal["minimize", "maximize"]:
        ...

    @abstractmethod
    def objective_function(self) -> float:
        ...

    # ------------------------------------------------------------------ hooks
    def _setup_adata(self, adata, background: bool = False) -> None:
        """Type check, keep a reference (no copy), clip the caller's X (signature_nmf.py:269-281).

        ``background`` (``fit`` only): the clipped copy the reference hands back in ``adata.X`` is made by a worker
        thread while the device already works from the raw matrix, which it clips itself on upload
        (``salnmf_upload_X_typed(clip=1)``: the same values); ``_finish_setup`` joins the thread and sets ``adata.X``
        -- before any host code reads it, and before ``fit`` returns at the latest.  The thread also touches the
        pages of the array the fitted exposures will be downloaded into."""
        type_checker("adata", adata, ANNDATA_TYPES)
        self.adata = adata
        if not background:
            self.adata.X = np.asarray(self.adata.X).clip(EPSILON)
            return
        raw = np.asarray(self.adata.X)
        box: dict[str, Any] = {}
        n_sigs = self.n_signatures

        def work():
            import time

            try:
                t0 = time.perf_counter()
                box["X"] = raw.clip(EPSILON)
                t1 = time.perf_counter()
                if raw.ndim == 2 and isinstance(n_sigs, int) and n_sigs > 0:
                    out = np.empty((raw.shape[0], n_sigs), dtype=np.float64)
                    np.copyto(out, 0.0)  # first touch here (the GIL is released), not inside the download at the end of the fit
                    box["H"] = out
                box["seconds"] = (t0, t1, time.perf_counter())  # (development aid: tools/time_fit_custom.py)
            except BaseException as exc:  # re-raised by _finish_setup
                box["error"] = exc

        self._raw_X = raw
        self._setup_box = box
        self._setup_thread = threading.Thread(target=work, name="salnmf-setup", daemon=True)
        self._setup_thread.start()

    def _finish_setup(self) -> None:
        """Join the worker of ``_setup_adata(background=True)``: ``adata.X`` becomes the clipped matrix."""
        thread = getattr(self, "_setup_thread", None)
        if thread is None:
            return
        thread.join()
        box = self._setup_box
        self._setup_thread = None
        raw, self._raw_X = self._raw_X, None
        if "error" in box:
            raise box["error"]
        self.adata.X = box["X"]
        self._exposures_out = box.get("H")
        # (adata.X just let go of the caller's unclipped matrix; if that was its last reference, returning 77 MB (c2) to the
        # system costs ~5 ms here -- which is why fit() calls this while it waits for the device)
        del raw

    @abstractmethod
    def _initialize(self, given_parameters=None, init_kwargs=None) -> None:
        ...

    @abstractmethod
    def _setup_fitting_parameters(self, fitting_kwargs=None) -> None:
        ...

    @abstractmethod
    def _update_parameters(self, given_parameters=None) -> None:
        ...

    # device-side counterparts used by the resident loop
    @abstractmethod
    def _device_steps(self, n_steps: int, given_parameters) -> None:
        ...

    @abstractmethod
    def _device_objective(self) -> float:
        ...

    def _device_weights(self):
        return None, None

    # ------------------------------------------------------------------ host <-> device state
    def _upload_X(self, e) -> None:
        """X to the device: the raw matrix, clipped there, while the background clip of ``fit`` is still running."""
        raw = getattr(self, "_raw_X", None)
        if raw is not None:
            e.upload_X(raw, clip=True)
        else:
            e.upload_X(np.ascontiguousarray(self.adata.X, dtype=np.float64))

    def _host_state(self):
        raw = getattr(self, "_raw_X", None)
        X = raw if raw is not None else np.ascontiguousarray(self.adata.X, dtype=np.float64)
        W = np.ascontiguousarray(self.asignatures.X, dtype=np.float64)
        H = None  # resident on the device only (fit() after a device-side initialisation)
        if "exposures" in self.adata.obsm or "H" not in self._resident:
            H = np.ascontiguousarray(self.adata.obsm["exposures"], dtype=np.float64)
        return X, W, H

    def _ensure_engine(self, N: int, V: int, K: int):
        """The engine of this problem shape (created on first use), its communicator attached when ``distributed``."""
        e = self._engine
        if e is None or (e.N, e.V, e.K, e.device) != (N, V, K, self.device):
            if e is not None:
                e.close()
            e = self._engine = Engine(N, V, K, device=self.device)
            if self.precision != "f64":
                e.set_precision(self.precision)
            self._comm_attached = False
            self._resident = set()
        if self.distributed and not self._comm_attached:
            from ..distributed import attach_communicator, attach_peer_exchange

            attach_communicator(e)
            # the K x V all-reduce of every step by peer stores where the GPUs of the node can map each other's
            # memory (include/salnmf.h: salnmf_p2p_*); otherwise every rank stays on the RCCL all-reduce
            attach_peer_exchange(e, required=False)
            self._comm_attached = True
            self._w_broadcast_due = True
        return e

    def _sync_to_device(self) -> None:
        X, W, H = self._host_state()
        N, V = X.shape
        K = W.shape[0]
        e = self._ensure_engine(N, V, K)
        if self.distributed and getattr(self, "_w_broadcast_due", False):
            from ..distributed import broadcast_from_rank0

            W = broadcast_from_rank0(W)  # every rank must start from bit-identical signatures
            self._w_broadcast_due = False
        # what a device-side initialisation left resident is not uploaded again (once)
        if "X" not in self._resident:
            self._upload_X(e)
        e.upload_W(W)
        if "H" not in self._resident:
            e.upload_H(H)
            self._initial_exposures_uploaded = True  # (fit: their host copy is let go of during the first device wait)
        self._resident = set()
        e.set_weights(*self._device_weights())

    def _sync_from_device(self) -> None:
        self.asignatures.X = self._engine.download_W()
        out = getattr(self, "_exposures_out", None)  # (fit: an array whose pages are touched already)
        self._exposures_out = None
        if out is not None and out.shape != (self._engine.N, self._engine.K):
            out = None
        self.adata.obsm["exposures"] = self._engine.download_H(out)

    # asynchronous objectives (fit): models whose objective can be queued without a host round trip override these
    def _device_can_queue(self) -> bool:
        """Whether ``fit`` may use the queued loop (the hooks below)."""
        return False

    def _device_objective_async(self, slot: int) -> bool:
        return False

    def _device_objectives_read(self, first: int, count: int):
        raise NotImplementedError

    def _device_steps_keep(self, n_steps: int, given_parameters) -> bool:
        """Queue ``n_steps`` updates that can be undone by ``_device_rollback`` (False: not supported)."""
        return False

    def _device_objective_and_steps(self, slot: int, n_steps: int, given_parameters, keep: bool) -> bool:
        """Queue the objective of the current state into ``slot``, then ``n_steps`` (>= 0) updates -- undoable ones if
        ``keep``.  Returns whether the updates were queued (False: ``keep`` asked for and not supported).  Models whose
        engine can evaluate the objective inside the first update override this (``KLNMF``)."""
        self._device_objective_async(slot)
        if n_steps == 0:
            return True
        if keep:
            return self._device_steps_keep(n_steps, given_parameters)
        self._device_steps(n_steps, given_parameters)
        return True

    def _device_rollback(self) -> None:
        raise NotImplementedError

    def _n_obs_total(self) -> int:
        """Samples over all shards (``adata`` is this rank's shard when ``distributed``), else ``adata.n_obs``."""
        if self.distributed and self._engine is not None and self._comm_attached:
            return int(self._engine.comm_info()[2])
        return int(self.adata.n_obs)

    @staticmethod
    def _n_given(given_parameters) -> int:
        if given_parameters and "asignatures" in given_parameters:
            return given_parameters["asignatures"].n_obs
        return 0

    # ------------------------------------------------------------------ fit (signature_nmf.py:315-385)
    def fit(
        self,
        adata,
        given_parameters: dict[str, Any] | None = None,
        init_kwargs: dict[str, Any] | None = None,
        fitting_kwargs: dict[str, Any] | None = None,
        history: bool = True,
        verbose: Literal[0, 1] = 0,
        verbosity_freq: int = 1000,
    ) -> "SignatureNMF":
        self._setup_adata(adata, background=self._background_setup)
        self._initial_exposures_uploaded = False
        self._initial_exposures_backup = None
        try:
            # inside fit() a device-side initialisation leaves the exposures on the device only: they come back once,
            # with the fitted ones (a 40 MB host array and its page faults less at c2)
            self._defer_exposures = True
            try:
                self._initialize(given_parameters, init_kwargs)
            finally:
                self._defer_exposures = False
            self._setup_fitting_parameters(fitting_kwargs)
            self._sync_to_device()
            self._fit_running = True  # (models may let the device run ahead between the blocks of one fit)
            try:
                if not verbose and self._device_can_queue():
                    of_values, n_iteration = self._fit_loop_queued(given_parameters)
                else:
                    of_values, n_iteration = self._fit_loop_blocking(given_parameters, verbose, verbosity_freq)
            finally:
                self._fit_running = False
        except BaseException:
            # the queued loop drops the host copy of the initial exposures while the device works (they come back fitted
            # at the end): a fit that does not get there hands the caller's AnnData back as the reference would leave it
            # after its initialisation (initialize.py:254), not without exposures
            if self._initial_exposures_backup is not None and "exposures" not in self.adata.obsm:
                self.adata.obsm["exposures"] = self._initial_exposures_backup
            raise
        finally:
            self._initial_exposures_backup = None
            self._finish_setup()  # adata.X is the clipped matrix from here on (signature_nmf.py:281)
        self._sync_from_device()
        self.n_iterations_ = n_iteration
        if history:
            self.history["objective_function"] = of_values[1:]
        return self

    def _next_stop(self, n_iteration: int) -> int:
        """The next iteration at which the reference would look at the model: a convergence test or the iteration cap
        (the reference tests ``n_iteration >= max_iterations`` after every single update: the cap is the next iteration
        at the latest, and ``max_iterations <= 0`` still runs exactly one update)."""
        freq = self.conv_test_freq
        stop = (n_iteration // freq + 1) * freq
        stop = min(stop, max(self.max_iterations, n_iteration + 1))
        return max(stop, n_iteration + 1)

    def _fit_loop_blocking(self, given_parameters, verbose, verbosity_freq):
        """The loop of signature_nmf.py:358-385 with one host round trip per objective (verbose fits, and models whose
        objective cannot be queued)."""
        of_values = [self._device_objective()]
        n_iteration = 0
        converged = False
        freq = self.conv_test_freq
        while not converged:
            # run up to the next iteration at which the reference would look at the model:
            # a convergence test, the iteration cap, or a verbosity print
            if verbose and (n_iteration + 1) % verbosity_freq == 0:
                # printed before the update of that iteration, with the latest known objective
                print(f"iteration: {n_iteration + 1}; objective: {of_values[-1]:.2f}")
            stop = self._next_stop(n_iteration)
            if verbose:
                next_print = ((n_iteration + 1) // verbosity_freq + 1) * verbosity_freq
                stop = min(stop, next_print - 1)
            stop = max(stop, n_iteration + 1)
            self._device_steps(stop - n_iteration, given_parameters)
            n_iteration = stop

            if n_iteration % freq == 0:
                prev = of_values[-1]
                of_values.append(self._device_objective())
                rel_change = np.abs(prev - of_values[-1]) / np.abs(prev)
                converged = bool(rel_change < self.tol and n_iteration >= self.min_iterations)
            converged |= n_iteration >= self.max_iterations
        return of_values, n_iteration

    def _fit_loop_queued(self, given_parameters):
        """The same loop with the host out of the way (same stopping iteration, objectives equal to rounding).

        * An objective is queued into the device's ring of slots TOGETHER with the updates that follow it
          (``_device_objective_and_steps``): the first of those updates forms ``H W`` of exactly the state the objective is
          about, so the engine evaluates the divergence inside that launch instead of in a forward pass of its own.
        * Before ``min_iterations`` the convergence test cannot stop the fit (signature_nmf.py:373-380): nothing is read;
          the values come back in one go when a decision first needs them.
        * From ``min_iterations`` on every test needs its objective: the block behind it is queued as updates that keep
          the state they start from (no copy), the read's round trip hides behind them, and if the test says
          "converged" the block is rolled back.
        The last objective of a fit (iteration cap reached) has no update behind it and is a forward pass."""
        from .. import _lib

        ring = _lib.OBJECTIVE_SLOTS
        freq = self.conv_test_freq
        of_values: list[float] = []
        first_slot, pending = 0, 0  # slots [first_slot, first_slot + pending) hold queued, unread objectives

        def read_pending():
            nonlocal first_slot, pending
            # The host is about to wait for the device: the moment to finish the background setup.  Letting go of the
            # caller's unclipped matrix returns 77 MB (c2) to the system, 5 ms during which the interpreter lock is held
            # by whoever frees it -- here it falls into a wait that does not need the interpreter.
            if pending:
                self._finish_setup()
                # likewise the host copy of the INITIAL exposures (40 MB at c2): the device has them, and fit() replaces
                # adata.obsm["exposures"] with the fitted ones at its end
                if getattr(self, "_initial_exposures_uploaded", False):
                    self._initial_exposures_uploaded = False
                    # (taken out of the caller's AnnData, kept by reference: fit() puts it back if the fit does not
                    # complete -- an engine error or an interrupt must not leave the object without exposures)
                    self._initial_exposures_backup = self.adata.obsm.pop("exposures", None)
            while pending:
                n = min(pending, ring - first_slot)
                of_values.extend(float(v) for v in self._device_objectives_read(first_slot, n))
                first_slot = (first_slot + n) % ring
                pending -= n

        n_iteration = 0  # the iteration the decisions below are about
        n_queued = 0     # updates launched so far (ahead of n_iteration by one block at most)
        while True:
            check = n_iteration % freq == 0  # (iteration 0: the initial objective)
            last = n_iteration > 0 and n_iteration >= self.max_iterations
            if check:
                decide = not last and n_iteration > 0 and n_iteration >= self.min_iterations
                nxt = n_iteration if last else self._next_stop(n_iteration)
                if pending == ring:
                    read_pending()
                queued = self._device_objective_and_steps((first_slot + pending) % ring, nxt - n_iteration, given_parameters, decide)
                pending += 1
                if queued:
                    n_queued = nxt
                if decide:
                    read_pending()
                    prev, cur = of_values[-2], of_values[-1]
                    if np.abs(prev - cur) / np.abs(prev) < self.tol:
                        if queued:
                            self._device_rollback()
                        break
            if last:
                break
            stop = self._next_stop(n_iteration)
            if n_queued < stop:
                self._device_steps(stop - n_queued, given_parameters)
                n_queued = stop
            n_iteration = stop
        read_pending()
        return of_values, n_iteration

    # ------------------------------------------------------------------ out of scope
    def _out_of_scope(self, *args, **kwargs):
        raise NotImplementedError(
            "plotting / post-hoc analysis helpers are outside the scope of salamander_amd "
            "(SURVEY.md section 2); use the reference package on the fitted AnnData objects."
        )

    reorder = plot_history = plot_signatures = plot_exposures = _out_of_scope
    plot_correlation = plot_embeddings = compute_correlation = correlation = _out_of_scope
