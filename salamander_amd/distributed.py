"""Sample-axis sharding across GPUs: one process per GPU, one engine per process.

The path shards naturally: every sample column is independent
given W; the only cross-sample quantity is the ``(K, V)`` numerator ``aux @ H.T`` of the W
update.  Each rank owns a contiguous block of rows of ``X (N, V)`` and ``H (N, K)``; W is
replicated; per step there is exactly one all-reduce of ``K*V`` doubles, after which every
rank runs the identical W tail so W stays bit-identical.

Two ways to run the exchange:

* **in-engine RCCL** (the product path): :func:`attach_communicator` bootstraps an RCCL
  communicator inside the engine (rank 0 creates the id, ``torch.distributed`` broadcasts
  it); ``Engine.kl_step`` then enqueues ``kernel -> ncclAllReduce -> W tail`` on the
  engine's stream with no host synchronisation inside the step loop.
* **host collective** (:func:`host_collective_steps`): the step split at the exchange
  point with ``torch.distributed.all_reduce`` on the numerator buffer.  Backend agnostic
  (``gloo`` on CPU for tests, ``nccl`` = RCCL on GPUs); used to test the decomposition.
"""

from __future__ import annotations

import numpy as np

from . import _lib


def shard_bounds(n_samples: int, world_size: int, rank: int) -> tuple[int, int]:
    """Contiguous row block ``[start, stop)`` of rank ``rank``; sizes differ by at most one tile of 16.

    Blocks are multiples of 16 rows (the engine's tile) except the last, so no tile straddles ranks.
    """
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    tiles = -(-n_samples // 16)
    base, extra = divmod(tiles, world_size)
    t0 = rank * base + min(rank, extra)
    t1 = t0 + base + (1 if rank < extra else 0)
    return min(t0 * 16, n_samples), min(t1 * 16, n_samples)


def _dist():
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run).")
    return dist


def attach_communicator(engine, group=None) -> None:
    """Create the engine's RCCL communicator across all ranks of ``group``."""
    dist = _dist()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    box = [engine.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    engine.comm_init(box[0], world, rank)


def attach_peer_exchange(engine, n_samples_total: int | None = None, group=None, required: bool = True, timeout_ms: int | None = None) -> bool:
    """Connect the engines of all ranks of ``group`` (one node) for the peer-to-peer exchange of the small
    all-reduces (``include/salnmf.h``: ``salnmf_p2p_export`` / ``salnmf_p2p_connect``).  The IPC handles and the shard
    sizes travel through ``torch.distributed`` (any backend); the exchange itself never touches the host.

    ``required=False``: if any rank cannot export or map an inbox (no peer access between two of the GPUs, more than 8
    ranks, ...) every rank leaves the exchange off -- the engines then need their RCCL communicator -- and False is
    returned on all of them instead of an exception on some.  ``timeout_ms``: how long an exchange waits for a peer
    before it gives up (default 20 s)."""
    dist = _dist()
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    error = None
    try:
        handle = engine.p2p_export(world)
        if timeout_ms is not None:
            engine.set_p2p_timeout(timeout_ms)
    except RuntimeError as exc:
        if required:
            raise
        handle, error = None, str(exc)
    gathered = [None] * world
    dist.all_gather_object(gathered, (handle, engine.N), group=group)
    ok = all(h is not None for h, _ in gathered)
    if ok:
        total = sum(n for _, n in gathered) if n_samples_total is None else int(n_samples_total)
        try:
            engine.p2p_connect(rank, [h for h, _ in gathered], total)
        except RuntimeError as exc:
            if required:
                raise
            ok, error = False, str(exc)
    # nobody starts exchanging before every rank has mapped every inbox -- and all ranks agree on whether to use it
    flags = [None] * world
    dist.all_gather_object(flags, ok, group=group)
    if not all(flags):
        if ok:
            engine.set_p2p(False)
        if error is not None:
            import warnings

            warnings.warn(f"peer-to-peer exchange not available on rank {rank}: {error}")
        return False
    return True


def broadcast_from_rank0(value, group=None):
    """Every rank gets rank 0's copy of ``value`` -- an array or a tuple of arrays / scalars (bit-identical
    replicated parameters at the start of a sharded fit)."""
    dist = _dist()
    if dist.get_rank(group) == 0:
        box = [np.ascontiguousarray(value) if isinstance(value, np.ndarray) else value]
    else:
        box = [None]
    dist.broadcast_object_list(box, src=0, group=group)
    return box[0]


class _DevicePointer:
    """Expose an engine buffer through ``__cuda_array_interface__`` so torch can wrap it zero-copy."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {
            "shape": (n,),
            "typestr": "<f8",
            "data": (ptr, False),
            "version": 2,
        }


def numerator_tensor(engine):
    """torch view of the engine's local ``(K*V,)`` numerator buffer (device memory, no copy)."""
    import torch

    ptr = engine.device_ptr(_lib.BUF_G)
    return torch.as_tensor(_DevicePointer(ptr, engine.K * engine.V), device=f"cuda:{engine.device}")


def host_collective_steps(engine, n_steps: int, n_given: int = 0, group=None) -> None:
    """``n_steps`` joint steps with the exchange done by ``torch.distributed.all_reduce``.

    ``engine`` needs ``kl_step_partial()``, ``numerator()`` (a tensor the collective may
    reduce in place), ``kl_step_finish(n_given, clip_mode)`` and ``sync()``; the real
    :class:`~salamander_amd.engine.Engine` is adapted by :class:`HostCollectiveAdapter`.
    """
    dist = _dist()
    for _ in range(n_steps):
        engine.kl_step_partial()
        if n_given < engine.K:  # all signatures given: W is untouched, nothing to exchange
            buf = engine.numerator()
            if getattr(buf, "is_cuda", False) and dist.get_backend(group) == "gloo":
                # a host-side control plane (rehearsals of several ranks on one GPU): the reduction goes through host memory
                host = buf.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
                buf.copy_(host)
            else:
                dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            engine.after_collective()
            engine.kl_step_finish(n_given, _lib.CLIP_ALL)


class HostCollectiveAdapter:
    """Adapts :class:`Engine` to :func:`host_collective_steps` (stream hand-over around the collective)."""

    def __init__(self, engine):
        self.engine = engine
        self.K = engine.K

    def kl_step_partial(self):
        self.engine.kl_step_partial()

    def numerator(self):
        self.engine.sync()  # the collective runs on torch's stream, not the engine's
        return numerator_tensor(self.engine)

    def after_collective(self):
        import torch

        torch.cuda.synchronize(self.engine.device)

    def kl_step_finish(self, n_given, clip_mode):
        self.engine.kl_step_finish(n_given, clip_mode)

    def sync(self):
        self.engine.sync()
