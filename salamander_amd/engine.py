"""Python handle of one device engine (one GPU, one shard of the sample axis).

Thin: validates shapes/dtypes, hands plain pointers to the C ABI
(``include/salnmf.h``) and converts status codes into exceptions.  Arrays are in
AnnData's storage layout: ``X (N, V)``, ``H (N, K)``, ``W (K, V)``, float64, C order.
"""

from __future__ import annotations

import ctypes
from ctypes import POINTER, c_double

import numpy as np

from . import _lib

_D = POINTER(c_double)


def _as_c(a, shape, name):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != tuple(shape):
        raise ValueError(f"The shape of '{name}' has to be {tuple(shape)}.")
    return a


def _ptr(a):
    return a.ctypes.data_as(_D)


class Engine:
    """Device-resident state of one KL-NMF problem shard: X, W, H (+ weights)."""

    def __init__(self, n_samples: int, n_features: int, n_signatures: int, device: int = 0):
        self._lib = _lib.load()
        if self._lib.salnmf_device_count() < 1:
            raise _lib.EngineUnavailable(
                "no HIP device visible: salamander_amd runs on MI355X (gfx950) only and has no CPU fallback."
            )
        self.N, self.V, self.K = int(n_samples), int(n_features), int(n_signatures)
        self.device = int(device)
        handle = ctypes.c_void_p()
        _lib.check(self._lib.salnmf_create(self.device, self.V, self.N, self.K, ctypes.byref(handle)))
        self._handle = handle

    # -- lifetime
    @property
    def _h(self):
        """The C handle; a closed engine raises instead of handing NULL to the library."""
        h = getattr(self, "_handle", None)
        if not h:
            raise RuntimeError("salnmf: this Engine has been closed")
        return h

    def close(self):
        if getattr(self, "_handle", None):
            self._lib.salnmf_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- transfers
    def upload_X(self, X, clip: bool = False):
        """The count matrix; float32 / int32 / int64 / uint16 arrays go over as they are and become float64 on the
        device, anything else is converted to float64 on the host first."""
        X = np.asarray(X)
        code = _lib.DTYPE_CODES.get(X.dtype.name)
        if code is None:
            X, code = X.astype(np.float64), 0
        X = np.ascontiguousarray(X)
        if X.shape != (self.N, self.V):
            raise ValueError(f"The shape of 'X' has to be {(self.N, self.V)}.")
        _lib.check(self._lib.salnmf_upload_X_typed(self._h, X.ctypes.data_as(ctypes.c_void_p), code, int(bool(clip))))

    def upload_W(self, W):
        W = _as_c(W, (self.K, self.V), "W")
        _lib.check(self._lib.salnmf_upload_W(self._h, _ptr(W)))

    def upload_H(self, H):
        H = _as_c(H, (self.N, self.K), "H")
        _lib.check(self._lib.salnmf_upload_H(self._h, _ptr(H)))

    def set_H_scale(self, scale):
        """Read H as ``clip(H * scale[k], EPSILON)`` from now on (``normalize_WH`` + clip of an initialisation, applied by
        the first pass that rewrites H)."""
        scale = _as_c(scale, (self.K,), "scale")
        _lib.check(self._lib.salnmf_set_H_scale(self._h, _ptr(scale)))

    def set_weights(self, weights_kl=None, weights_lhalf=None):
        wk = None if weights_kl is None else _as_c(weights_kl, (self.N,), "weights_kl")
        wl = None if weights_lhalf is None else _as_c(weights_lhalf, (self.N,), "weights_lhalf")
        _lib.check(
            self._lib.salnmf_set_weights(self._h, None if wk is None else _ptr(wk), None if wl is None else _ptr(wl))
        )

    def download_W(self) -> np.ndarray:
        W = np.empty((self.K, self.V), dtype=np.float64)
        _lib.check(self._lib.salnmf_download_W(self._h, _ptr(W)))
        return W

    def download_H(self, out: np.ndarray | None = None) -> np.ndarray:
        """The exposures; ``out``: a C-contiguous float64 ``(N, K)`` array to fill (e.g. one whose pages were touched
        beforehand: a fresh 400 MB array costs more in page faults than its transfer)."""
        if out is None:
            out = np.empty((self.N, self.K), dtype=np.float64)
        elif out.shape != (self.N, self.K) or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError(f"'out' has to be a C-contiguous float64 array of shape {(self.N, self.K)}.")
        _lib.check(self._lib.salnmf_download_H(self._h, _ptr(out)))
        return out

    # -- KLNMF
    def kl_step(self, n_steps: int = 1, n_given: int = 0):
        _lib.check(self._lib.salnmf_kl_step(self._h, int(n_steps), int(n_given)))

    def set_precision(self, precision: str = "f64"):
        """``"f64"`` (default; the reference's arithmetic) or ``"f32"``: ``kl_step`` on the fp32 matrix cores, an opt-in
        fast mode with its own tolerance (``include/salnmf.h: salnmf_set_precision``)."""
        if precision not in _lib.PRECISIONS:
            raise ValueError(f"precision has to be one of {sorted(_lib.PRECISIONS)}.")
        _lib.check(self._lib.salnmf_set_precision(self._h, _lib.PRECISIONS[precision]))

    def set_lockstep(self, on: bool = True):
        """Signature-embedding solves in lockstep rounds (default from 2 048 samples on) or one workgroup per signature."""
        _lib.check(self._lib.salnmf_set_lockstep(self._h, int(bool(on))))

    def set_w_dma(self, on: bool = True):
        """W into the workgroups' LDS by LDS-DMA (default) or through registers (``include/salnmf.h: salnmf_set_w_dma``)."""
        _lib.check(self._lib.salnmf_set_w_dma(self._h, int(bool(on))))

    def set_mv_queued(self, on: bool = True):
        """MvNMF steps queued ahead of the host with the line-search decision on the device (default) or the classic form
        (``include/salnmf.h: salnmf_set_mv_queued``)."""
        _lib.check(self._lib.salnmf_set_mv_queued(self._h, int(bool(on))))

    def set_small_cohort_tiles(self, max_tiles: int = 8):
        """Up to ``max_tiles`` tiles of 16 samples ``kl_step`` runs as one workgroup, all steps of a call in one launch
        (``include/salnmf.h: salnmf_set_small_cohort_tiles``); 0 turns that off."""
        _lib.check(self._lib.salnmf_set_small_cohort_tiles(self._h, int(max_tiles)))

    def set_batched_sample_solves(self, on: bool = True):
        """Sample-embedding solves sixteen per wavefront on the MFMA units (default where the shape allows) or one wavefront per sample."""
        _lib.check(self._lib.salnmf_set_batched_sample_solves(self._h, int(bool(on))))

    def set_persistent(self, on: bool = True):
        """Run multi-step ``kl_step`` calls as one persistent launch (only in builds with ``SALNMF_WITH_PERSISTENT=1``:
        measured 10 % slower, so the default library does not carry that kernel) or as per-step launches (default)."""
        _lib.check(self._lib.salnmf_set_persistent(self._h, int(bool(on))))

    def update_H(self):
        _lib.check(self._lib.salnmf_update_H(self._h))

    def update_W(self, n_given: int = 0, clip_mode: int = _lib.CLIP_NON_GIVEN):
        _lib.check(self._lib.salnmf_update_W(self._h, int(n_given), int(clip_mode)))

    def objective(self) -> float:
        out = c_double()
        _lib.check(self._lib.salnmf_objective(self._h, ctypes.byref(out)))
        return out.value

    def objective_async(self, slot: int):
        """Queue the objective of the resident state into slot ``slot`` of the device ring (no host round trip)."""
        _lib.check(self._lib.salnmf_objective_async(self._h, int(slot)))

    def objective_read(self, first: int, count: int) -> np.ndarray:
        """The values of ``count`` slots from ``first`` on (blocks until everything queued has finished)."""
        out = np.empty(int(count), dtype=np.float64)
        _lib.check(self._lib.salnmf_objective_read(self._h, int(first), int(count), _ptr(out)))
        return out

    def kl_step_keep(self, n_steps: int = 1, n_given: int = 0):
        """``kl_step`` that keeps the state it starts from; :meth:`kl_rollback` returns to it."""
        _lib.check(self._lib.salnmf_kl_step_keep(self._h, int(n_steps), int(n_given)))

    def kl_step_objective(self, slot: int, n_steps: int, n_given: int = 0, keep: bool = False):
        """Objective of the resident state into ring slot ``slot``, then ``n_steps`` updates (``keep``: undoable); where
        the engine can, the objective is evaluated inside the first update's launch (``salnmf_kl_step_objective``)."""
        _lib.check(self._lib.salnmf_kl_step_objective(self._h, int(slot), int(n_steps), int(n_given), 1 if keep else 0))

    def kl_rollback(self):
        _lib.check(self._lib.salnmf_kl_rollback(self._h))

    def samplewise_kl(self) -> np.ndarray:
        out = np.empty(self.N, dtype=np.float64)
        _lib.check(self._lib.salnmf_samplewise_kl(self._h, _ptr(out)))
        return out

    def reconstruct(self) -> np.ndarray:
        out = np.empty((self.N, self.V), dtype=np.float64)
        _lib.check(self._lib.salnmf_reconstruct(self._h, _ptr(out)))
        return out

    # -- MvNMF
    def mv_step(self, n_steps: int, n_given: int, lam: float, delta: float, gamma: float) -> float:
        g = c_double(gamma)
        _lib.check(self._lib.salnmf_mv_step(self._h, int(n_steps), int(n_given), float(lam), float(delta), ctypes.byref(g)))
        return g.value

    def mv_step_objective(self, n_steps: int, n_given: int, lam: float, delta: float, gamma: float, more_follows: bool = False) -> tuple[float, float]:
        """``mv_step`` that also returns the objective of the state it leaves behind (the last line search's accepted
        value): ``(gamma, objective)``.  ``more_follows``: the caller expects to continue with another ``mv_step``; the
        engine then keeps the speculative first half of that step across the calls (any other call steps back first)."""
        g, f = c_double(gamma), c_double()
        _lib.check(self._lib.salnmf_mv_step_objective(self._h, int(n_steps), int(n_given), float(lam), float(delta), ctypes.byref(g), ctypes.byref(f),
                                                      1 if more_follows else 0))
        return g.value, f.value

    def mv_logdet(self, delta: float) -> float:
        """``volume_logdet`` (mvnmf.py:19-24) of the resident signatures."""
        out = c_double()
        _lib.check(self._lib.salnmf_mv_logdet(self._h, float(delta), ctypes.byref(out)))
        return out.value

    def mv_update_W_unconstrained(self, n_given: int, lam: float, delta: float) -> np.ndarray:
        """``update_W_unconstrained`` (mvnmf.py:37-66) from the resident state: ``(K, V)``; nothing resident changes."""
        out = np.empty((self.K, self.V), dtype=np.float64)
        _lib.check(self._lib.salnmf_mv_update_W_unconstrained(self._h, int(n_given), float(lam), float(delta), _ptr(out)))
        return out

    def mv_line_search(self, lam: float, delta: float, gamma: float, W_unconstrained) -> float:
        """``line_search`` (mvnmf.py:69-92) from the resident state with the given ``W_unconstrained (K, V)``; the accepted
        W and the rescaled H stay resident, the new gamma is returned."""
        Wu = _as_c(W_unconstrained, (self.K, self.V), "W_unconstrained")
        g = c_double(gamma)
        _lib.check(self._lib.salnmf_mv_line_search(self._h, float(lam), float(delta), ctypes.byref(g), _ptr(Wu)))
        return g.value

    def mv_update_W(self, n_given: int, lam: float, delta: float, gamma: float) -> float:
        g = c_double(gamma)
        _lib.check(self._lib.salnmf_mv_update_W(self._h, int(n_given), float(lam), float(delta), ctypes.byref(g)))
        return g.value

    def mv_objective(self, lam: float, delta: float) -> float:
        out = c_double()
        _lib.check(self._lib.salnmf_mv_objective(self._h, float(lam), float(delta), ctypes.byref(out)))
        return out.value

    # -- CorrNMF dense pieces (the embedding solves stay with the host layer)
    def corr_configure(self, dim_embeddings: int):
        _lib.check(self._lib.salnmf_corr_configure(self._h, int(dim_embeddings)))
        self.dim = int(dim_embeddings)

    def _corr_shape(self, which: int):
        return {
            _lib.CORR_SIGNATURE_SCALINGS: (self.K,),
            _lib.CORR_SAMPLE_SCALINGS: (self.N,),
            _lib.CORR_SIGNATURE_EMBEDDINGS: (self.K, self.dim),
            _lib.CORR_SAMPLE_EMBEDDINGS: (self.N, self.dim),
            _lib.CORR_AUX: (self.N, self.K),
        }[which]

    def corr_upload(self, which: int, values):
        a = _as_c(values, self._corr_shape(which), "values")
        _lib.check(self._lib.salnmf_corr_upload(self._h, int(which), _ptr(a)))

    def corr_download(self, which: int) -> np.ndarray:
        out = np.empty(self._corr_shape(which), dtype=np.float64)
        _lib.check(self._lib.salnmf_corr_download(self._h, int(which), _ptr(out)))
        return out

    def corr_update_sample_scalings(self):
        _lib.check(self._lib.salnmf_corr_update_sample_scalings(self._h))

    def corr_compute_exposures(self):
        _lib.check(self._lib.salnmf_corr_compute_exposures(self._h))

    def corr_compute_aux(self):
        _lib.check(self._lib.salnmf_corr_compute_aux(self._h))

    def corr_update_signature_scalings(self):
        _lib.check(self._lib.salnmf_corr_update_signature_scalings(self._h))

    def corr_update_signatures(self, n_given: int = 0):
        _lib.check(self._lib.salnmf_corr_update_signatures(self._h, int(n_given)))

    def corr_update_sample_embeddings(self, variance: float, maxiter: int = 3, return_status: bool = False):
        """One Newton-CG solve per sample on the device; optionally the per-sample SciPy-style status codes."""
        status = np.empty(self.N, dtype=np.int32) if return_status else None
        ptr = status.ctypes.data_as(POINTER(ctypes.c_int)) if return_status else None
        _lib.check(self._lib.salnmf_corr_update_sample_embeddings(self._h, float(variance), int(maxiter), ptr))
        return status

    @staticmethod
    def corr_update_sample_embeddings_multi(engines, variance: float, maxiter: int = 3, return_status: bool = False):
        """Joint solve of the sample embeddings shared by several modalities (one engine each)."""
        engines = list(engines)
        handles = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
        status = np.empty(engines[0].N, dtype=np.int32) if return_status else None
        ptr = status.ctypes.data_as(POINTER(ctypes.c_int)) if return_status else None
        _lib.check(engines[0]._lib.salnmf_corr_update_sample_embeddings_multi(handles, len(engines), float(variance), int(maxiter), ptr))
        return status

    def corr_update_signature_embeddings(self, variance: float, maxiter: int = 0, return_status: bool = False):
        """One Newton-CG solve per signature on the device (SciPy's default iteration limit when maxiter <= 0)."""
        status = np.empty(self.K, dtype=np.int32) if return_status else None
        ptr = status.ctypes.data_as(POINTER(ctypes.c_int)) if return_status else None
        _lib.check(self._lib.salnmf_corr_update_signature_embeddings(self._h, float(variance), int(maxiter), ptr))
        return status

    def corr_update_signature_embeddings_from(self, U_all, alpha_all, aux_all, variance: float, maxiter: int = 0, return_status: bool = False):
        """The signature solves on caller-gathered sample-side inputs of ALL shards (global sample order)."""
        n_all = int(np.shape(alpha_all)[0])
        U_all = _as_c(U_all, (n_all, self.dim), "U_all")
        alpha_all = _as_c(alpha_all, (n_all,), "alpha_all")
        aux_all = _as_c(aux_all, (n_all, self.K), "aux_all")
        status = np.empty(self.K, dtype=np.int32) if return_status else None
        ptr = status.ctypes.data_as(POINTER(ctypes.c_int)) if return_status else None
        _lib.check(
            self._lib.salnmf_corr_update_signature_embeddings_from(
                self._h, n_all, _ptr(U_all), _ptr(alpha_all), _ptr(aux_all), float(variance), int(maxiter), ptr
            )
        )
        return status

    def corr_embedding_sumsq(self):
        """(sum of squares of the signature embeddings, of the sample embeddings) of the resident state."""
        out = (c_double * 2)()
        _lib.check(self._lib.salnmf_corr_embedding_sumsq(self._h, out))
        return out[0], out[1]

    def corr_poisson_llh(self) -> float:
        out = c_double()
        _lib.check(self._lib.salnmf_corr_poisson_llh(self._h, ctypes.byref(out)))
        return out.value

    # -- initialisation on the device (row f3)
    def init_gram(self):
        """(X^T X (V, V), sum of X) of the resident X, over all shards."""
        G = np.empty((self.V, self.V), dtype=np.float64)
        total = c_double()
        _lib.check(self._lib.salnmf_init_gram(self._h, _ptr(G), ctypes.byref(total)))
        return G, total.value

    def init_project(self, B):
        """H <- X B^T; returns the squared norms of the positive and of the negative part of every column."""
        B = _as_c(B, (self.K, self.V), "B")
        out = np.empty(2 * self.K, dtype=np.float64)
        _lib.check(self._lib.salnmf_init_project(self._h, _ptr(B), _ptr(out)))
        return out[: self.K].copy(), out[self.K :].copy()

    def init_finish(self, scale, take_neg, post, zero_below: float, fill: float):
        scale = _as_c(scale, (self.K,), "scale")
        post = _as_c(post, (self.K,), "post")
        neg = np.ascontiguousarray(take_neg, dtype=np.int32)
        if neg.shape != (self.K,):
            raise ValueError(f"The shape of 'take_neg' has to be {(self.K,)}.")
        _lib.check(self._lib.salnmf_init_finish(self._h, _ptr(scale), neg.ctypes.data_as(POINTER(ctypes.c_int)), _ptr(post), float(zero_below), float(fill)))

    def init_separable(self, n_select: int, return_norms: bool = False):
        """Sample indices chosen by the successive projection of ``separableNMF`` on the resident X, in selection order;
        ``return_norms``: also the winning squared norm of every round."""
        out = np.empty(int(n_select), dtype=np.int64)
        norms = np.empty(int(n_select), dtype=np.float64)
        _lib.check(self._lib.salnmf_init_separable(self._h, int(n_select), out.ctypes.data_as(POINTER(ctypes.c_int64)), _ptr(norms)))
        return (out, norms) if return_norms else out

    def init_flat(self, post):
        post = _as_c(post, (self.K,), "post")
        _lib.check(self._lib.salnmf_init_flat(self._h, _ptr(post)))

    # -- multi-GPU
    @staticmethod
    def comm_unique_id() -> bytes:
        buf = ctypes.create_string_buffer(_lib.UNIQUE_ID_BYTES)
        _lib.check(_lib.load().salnmf_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, n_ranks: int, rank: int):
        if len(unique_id) != _lib.UNIQUE_ID_BYTES:
            raise ValueError("unique_id has the wrong length")
        _lib.check(self._lib.salnmf_comm_init(self._h, unique_id, int(n_ranks), int(rank)))

    def comm_info(self):
        """(n_ranks, rank, n_samples over all shards)."""
        n, r, t = ctypes.c_int(), ctypes.c_int(), ctypes.c_int64()
        _lib.check(self._lib.salnmf_comm_info(self._h, ctypes.byref(n), ctypes.byref(r), ctypes.byref(t)))
        return n.value, r.value, t.value

    def comm_observed(self) -> dict:
        """What the exchange layers themselves report (``include/salnmf.h: salnmf_comm_observed``): the RCCL communicator's own
        rank count / rank / device (-1 without one), the peer-to-peer exchange's rank count and mapped inboxes, this engine's
        device and PCI bus id."""
        ints = [ctypes.c_int() for _ in range(6)]
        peers = (ctypes.c_int * 8)()
        bus = ctypes.create_string_buffer(32)
        _lib.check(self._lib.salnmf_comm_observed(self._h, *[ctypes.byref(i) for i in ints[:5]], peers, ctypes.byref(ints[5]), bus))
        keys = ("rccl_nranks", "rccl_rank", "rccl_device", "p2p_nranks", "p2p_inboxes_mapped", "device")
        out = {k: i.value for k, i in zip(keys, ints)}
        out["p2p_peer_devices"] = [int(v) for v in peers][: max(out["p2p_nranks"], 0)]
        out["pci_bus_id"] = bus.value.decode()
        return out

    def p2p_export(self, n_ranks: int, max_count: int | None = None) -> bytes:
        """Allocate this engine's inbox of the peer-to-peer exchange (``include/salnmf.h``) and return its IPC handle.

        ``max_count`` defaults to the inbox limit of 16 384 doubles (4 MB of inbox at 8 ranks): the numerator with the MvNMF
        sums (K*V + K + 2), the Gram matrix of the device-side initialisation (96*96 + 1) and, for small ``dim_embeddings``,
        the evaluation records of CorrNMF's lockstep signature solves (K * (66 + dim^2)) all fit; larger all-reduces (wide or
        many-signature engines, dim 40) go through the RCCL communicator."""
        if max_count is None:
            max_count = _lib.P2P_MAX_COUNT
        buf = ctypes.create_string_buffer(_lib.P2P_HANDLE_BYTES)
        _lib.check(self._lib.salnmf_p2p_export(self._h, int(n_ranks), int(max_count), buf))
        return buf.raw

    def p2p_connect(self, rank: int, handles, n_samples_total: int):
        """Map the peers' inboxes; ``handles``: one ``p2p_export`` result per rank, in rank order."""
        handles = list(handles)
        if any(len(h) != _lib.P2P_HANDLE_BYTES for h in handles):
            raise ValueError("a handle has the wrong length")
        _lib.check(self._lib.salnmf_p2p_connect(self._h, int(rank), len(handles), b"".join(handles), int(n_samples_total)))

    def set_p2p_timeout(self, timeout_ms: int):
        """How long an exchange waits for a peer before it gives up (default 20 s)."""
        _lib.check(self._lib.salnmf_set_p2p_timeout_ms(self._h, int(timeout_ms)))

    def set_p2p(self, on: bool):
        _lib.check(self._lib.salnmf_set_p2p(self._h, int(bool(on))))

    def kl_step_partial(self):
        _lib.check(self._lib.salnmf_kl_step_partial(self._h))

    def kl_step_finish(self, n_given: int = 0, clip_mode: int = _lib.CLIP_ALL):
        _lib.check(self._lib.salnmf_kl_step_finish(self._h, int(n_given), int(clip_mode)))

    def device_ptr(self, which: int) -> int:
        return int(self._lib.salnmf_device_ptr(self._h, int(which)) or 0)

    def sync(self):
        _lib.check(self._lib.salnmf_sync(self._h))

    # -- measurement
    def profile_kl_steps(self, n_steps: int, n_given: int = 0, sample_stride: int = 8):
        """Run ``n_steps`` joint steps with HIP events around every ``sample_stride``-th step's launches.

        Returns (total_ms, fused_kernel_avg_ms, tail_avg_ms)."""
        t, f, w = c_double(), c_double(), c_double()
        _lib.check(
            self._lib.salnmf_profile_kl_steps(
                self._h, int(n_steps), int(n_given), int(sample_stride), ctypes.byref(t), ctypes.byref(f), ctypes.byref(w)
            )
        )
        return t.value, f.value, w.value

    def profile_sharded_steps(self, n_steps: int, n_given: int = 0) -> dict:
        """Per-phase timeline of this rank's sharded joint step in microseconds (``include/salnmf.h:
        salnmf_profile_sharded_steps``): HIP events around the two launches, in-kernel stamps around the exchange."""
        out = (c_double * 9)()
        _lib.check(self._lib.salnmf_profile_sharded_steps(self._h, int(n_steps), int(n_given), out))
        keys = ("step", "fused_pass", "tail_exchange_launch", "local_slab_reduce", "peer_stores_and_flags", "wait_for_peer_flags",
                "read_and_sum_peer_rows", "w_row_finish", "longest_flag_wait")
        return {k: float(v) for k, v in zip(keys, out)}

    def profile_objective(self, n_calls: int) -> float:
        a = c_double()
        _lib.check(self._lib.salnmf_profile_objective(self._h, int(n_calls), ctypes.byref(a)))
        return a.value

    def profile_reconstruct(self, n_calls: int) -> float:
        """Average duration (ms) of the forward kernel alone (H @ W into a scratch buffer)."""
        a = c_double()
        _lib.check(self._lib.salnmf_profile_reconstruct(self._h, int(n_calls), ctypes.byref(a)))
        return a.value
