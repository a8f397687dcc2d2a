"""ctypes binding of the C ABI in ``include/salnmf.h`` (the drop-in boundary).

The shared library is built in-tree by ``__graft_entry__.build()`` /
``salamander_amd/csrc/build.sh``.  There is **no** CPU fallback: if the library is
missing or no gfx950 device is present, every compute entry point raises.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_int, c_int64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SALNMF_LIB: a development hook of this loader -- tools/ A/B runs of two builds of the library in one gpurun call)
LIB_PATH = os.environ.get("SALNMF_LIB") or os.path.join(_HERE, "lib", "libsalnmf.so")

UNIQUE_ID_BYTES = 128
P2P_HANDLE_BYTES = 64
P2P_MAX_COUNT = 16384  # salnmf_p2p_kernels.h: P2P_MAX_WG * P2P_BLOCK doubles per exchange
PRECISIONS = {"f64": 0, "f32": 1}
DTYPE_CODES = {"float64": 0, "float32": 1, "int32": 2, "int64": 3, "uint16": 4}  # SALNMF_F64 ...
CLIP_ALL = 0
CLIP_NON_GIVEN = 1
BUF_G, BUF_W, BUF_H, BUF_X, BUF_OBJ, BUF_RED = range(6)
CORR_SIGNATURE_SCALINGS, CORR_SAMPLE_SCALINGS, CORR_SIGNATURE_EMBEDDINGS, CORR_SAMPLE_EMBEDDINGS, CORR_AUX = range(5)

# every symbol include/salnmf.h declares: name -> (restype, argtypes)
_P = c_void_p
_D = POINTER(c_double)
SIGNATURES = {
    "salnmf_last_error": (c_char_p, []),
    "salnmf_version": (c_int, []),
    "salnmf_build_flags": (c_int, []),
    "salnmf_device_count": (c_int, []),
    "salnmf_create": (c_int, [c_int, c_int, c_int64, c_int, POINTER(_P)]),
    "salnmf_destroy": (None, [_P]),
    "salnmf_upload_X": (c_int, [_P, _D, c_int]),
    "salnmf_upload_X_typed": (c_int, [_P, c_void_p, c_int, c_int]),
    "salnmf_upload_W": (c_int, [_P, _D]),
    "salnmf_upload_H": (c_int, [_P, _D]),
    "salnmf_set_weights": (c_int, [_P, _D, _D]),
    "salnmf_set_H_scale": (c_int, [_P, _D]),
    "salnmf_download_W": (c_int, [_P, _D]),
    "salnmf_download_H": (c_int, [_P, _D]),
    "salnmf_kl_step": (c_int, [_P, c_int, c_int]),
    "salnmf_set_persistent": (c_int, [_P, c_int]),
    "salnmf_set_lockstep": (c_int, [_P, c_int]),
    "salnmf_set_batched_sample_solves": (c_int, [_P, c_int]),
    "salnmf_set_small_cohort_tiles": (c_int, [_P, c_int]),
    "salnmf_set_mv_queued": (c_int, [_P, c_int]),
    "salnmf_set_w_dma": (c_int, [_P, c_int]),
    "salnmf_set_precision": (c_int, [_P, c_int]),
    "salnmf_update_H": (c_int, [_P]),
    "salnmf_update_W": (c_int, [_P, c_int, c_int]),
    "salnmf_objective": (c_int, [_P, _D]),
    "salnmf_objective_async": (c_int, [_P, c_int]),
    "salnmf_objective_read": (c_int, [_P, c_int, c_int, _D]),
    "salnmf_kl_step_keep": (c_int, [_P, c_int, c_int]),
    "salnmf_kl_step_objective": (c_int, [_P, c_int, c_int, c_int, c_int]),
    "salnmf_kl_rollback": (c_int, [_P]),
    "salnmf_samplewise_kl": (c_int, [_P, _D]),
    "salnmf_reconstruct": (c_int, [_P, _D]),
    "salnmf_mv_step": (c_int, [_P, c_int, c_int, c_double, c_double, _D]),
    "salnmf_mv_logdet": (c_int, [_P, c_double, _D]),
    "salnmf_mv_update_W_unconstrained": (c_int, [_P, c_int, c_double, c_double, _D]),
    "salnmf_mv_line_search": (c_int, [_P, c_double, c_double, _D, _D]),
    "salnmf_mv_step_objective": (c_int, [_P, c_int, c_int, c_double, c_double, _D, _D, c_int]),
    "salnmf_mv_update_W": (c_int, [_P, c_int, c_double, c_double, _D]),
    "salnmf_mv_objective": (c_int, [_P, c_double, c_double, _D]),
    "salnmf_corr_configure": (c_int, [_P, c_int]),
    "salnmf_corr_upload": (c_int, [_P, c_int, _D]),
    "salnmf_corr_download": (c_int, [_P, c_int, _D]),
    "salnmf_corr_update_sample_scalings": (c_int, [_P]),
    "salnmf_corr_compute_exposures": (c_int, [_P]),
    "salnmf_corr_compute_aux": (c_int, [_P]),
    "salnmf_corr_update_signature_scalings": (c_int, [_P]),
    "salnmf_corr_update_signatures": (c_int, [_P, c_int]),
    "salnmf_corr_update_sample_embeddings": (c_int, [_P, c_double, c_int, POINTER(c_int)]),
    "salnmf_corr_update_sample_embeddings_multi": (c_int, [POINTER(_P), c_int, c_double, c_int, POINTER(c_int)]),
    "salnmf_corr_update_signature_embeddings": (c_int, [_P, c_double, c_int, POINTER(c_int)]),
    "salnmf_corr_update_signature_embeddings_from": (c_int, [_P, c_int64, _D, _D, _D, c_double, c_int, POINTER(c_int)]),
    "salnmf_corr_embedding_sumsq": (c_int, [_P, _D]),
    "salnmf_corr_poisson_llh": (c_int, [_P, _D]),
    "salnmf_init_gram": (c_int, [_P, _D, _D]),
    "salnmf_init_project": (c_int, [_P, _D, _D]),
    "salnmf_init_finish": (c_int, [_P, _D, POINTER(c_int), _D, c_double, c_double]),
    "salnmf_init_flat": (c_int, [_P, _D]),
    "salnmf_init_separable": (c_int, [_P, c_int, POINTER(c_int64), _D]),
    "salnmf_comm_unique_id": (c_int, [ctypes.c_char_p]),
    "salnmf_comm_init": (c_int, [_P, ctypes.c_char_p, c_int, c_int]),
    "salnmf_comm_info": (c_int, [_P, POINTER(c_int), POINTER(c_int), POINTER(c_int64)]),
    "salnmf_comm_observed": (c_int, [_P, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int), ctypes.c_char_p]),
    "salnmf_p2p_export": (c_int, [_P, c_int, c_int64, ctypes.c_char_p]),
    "salnmf_set_p2p_timeout_ms": (c_int, [_P, c_int64]),
    "salnmf_p2p_connect": (c_int, [_P, c_int, c_int, ctypes.c_char_p, c_int64]),
    "salnmf_set_p2p": (c_int, [_P, c_int]),
    "salnmf_kl_step_partial": (c_int, [_P]),
    "salnmf_kl_step_finish": (c_int, [_P, c_int, c_int]),
    "salnmf_device_ptr": (c_void_p, [_P, c_int]),
    "salnmf_stream": (c_void_p, [_P]),
    "salnmf_sync": (c_int, [_P]),
    "salnmf_profile_kl_steps": (c_int, [_P, c_int, c_int, c_int, _D, _D, _D]),
    "salnmf_profile_objective": (c_int, [_P, c_int, _D]),
    "salnmf_profile_sharded_steps": (c_int, [_P, c_int, c_int, POINTER(c_double)]),
    "salnmf_profile_reconstruct": (c_int, [_P, c_int, _D]),
}

OBJECTIVE_SLOTS = 256  # SALNMF_OBJECTIVE_SLOTS
BUILD_PERSISTENT = 1  # salnmf_build_flags(): the library carries the persistent multi-step kernel

_lib = None
# explicit selector (no environment switch): whether load() opens PyTorch-ROCm's bundled HIP runtime first
SHARE_TORCH_RUNTIME = True


class EngineUnavailable(RuntimeError):
    """The HIP extension (or a gfx950 device) is missing.  There is no fallback."""


def _share_torch_hip_runtime() -> None:
    """Make this library share PyTorch-ROCm's bundled HIP runtime when torch is installed.

    The torch wheel ships its own ``libamdhip64.so`` (SONAME ``libamdhip64.so.7``) and asks for it by the
    unversioned file name; ``libsalnmf.so`` asks for the SONAME.  Whichever is loaded first decides: with torch's
    copy first both resolve to it; with the system copy first torch later loads a second runtime, and of two HIP
    runtimes in one process the one that initialises second sees no device.  So torch's ``libamdhip64.so`` is
    opened first, by path and without importing torch (20 ms instead of seconds); a later ``import torch`` finds
    it loaded.  Only the HIP runtime is preloaded.  RCCL is not a load-time dependency of ``libsalnmf.so`` at all: the engine
    binds it when a communicator is first needed (``rccl_bind`` in ``csrc/salnmf.hip``: ``dlopen(RTLD_NOLOAD)`` by
    SONAME), and by then every ``torch.distributed`` program has imported torch, so the engine and torch's NCCL backend
    share torch's ``librccl.so`` whatever the import order; a process without torch gets the system library.  (Opening
    torch's ``librccl.so`` by hand from a process that never initialises torch aborted in that library's static
    destructors at exit -- nothing here does that any more.)  :func:`mapped_runtime_libraries` reports what is mapped;
    the tests assert one HIP runtime and at most one RCCL for both import orders.
    ``salamander_amd._lib.SHARE_TORCH_RUNTIME = False`` (set before the first engine is created) disables the preload;
    without torch the system libraries are used."""
    import importlib.util
    import sys

    if "torch" in sys.modules or not SHARE_TORCH_RUNTIME:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass  # an incomplete torch install: fall back to the system runtime


def load():
    """Load ``libsalnmf.so`` and declare every prototype.  Loading needs no GPU."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  salamander_amd has no CPU fallback."
        )
    _share_torch_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def mapped_runtime_libraries() -> dict:
    """Paths of the HIP runtime and RCCL copies mapped into this process: ``{"libamdhip64": [...], "librccl": [...]}``."""
    import re

    found = {"libamdhip64": set(), "librccl": set()}
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                m = re.search(r"(/\S*/(libamdhip64|librccl)\.so\S*)", line)
                if m:
                    found[m.group(2)].add(os.path.realpath(m.group(1)))
    except OSError:
        pass
    return {k: sorted(v) for k, v in found.items()}


def last_error() -> str:
    msg = load().salnmf_last_error()
    return msg.decode() if msg else ""


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError(f"salnmf: {last_error()}")
