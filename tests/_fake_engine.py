"""An oracle-backed stand-in for ``salamander_amd.engine.Engine`` -- TESTS ONLY.

Lets the CPU suite exercise the host logic (fit loop, hooks, sharded driver) without a
GPU.  It is never importable from the product package.
"""

import numpy as np

from oracle import klnmf_oracle as orc


class FakeEngine:
    instances = 0

    def __init__(self, n_samples, n_features, n_signatures, device=0):
        self.N, self.V, self.K, self.device = n_samples, n_features, n_signatures, device
        self.wkl = self.wlh = None
        self.steps_log = []
        FakeEngine.instances += 1

    def close(self):
        pass

    def upload_X(self, X, clip=False):
        self.X = np.array(X, dtype=float)
        if clip:
            self.X = self.X.clip(orc.EPSILON)

    def upload_W(self, W):
        self.W = np.array(W, dtype=float)

    def upload_H(self, H):
        self.H = np.array(H, dtype=float)

    def set_H_scale(self, scale):
        self.H = np.clip(self.H * np.asarray(scale)[None, :], orc.EPSILON, None)

    def set_weights(self, wkl=None, wlh=None):
        self.wkl, self.wlh = wkl, wlh

    def download_W(self):
        return self.W.copy()

    def download_H(self, out=None):
        if out is None:
            return self.H.copy()
        out[...] = self.H
        return out

    def kl_step(self, n_steps=1, n_given=0):
        self.steps_log.append(n_steps)
        for _ in range(n_steps):
            W, H = orc.update_WH(self.X.T, self.W.T, self.H.T, self.wkl, self.wlh, n_given)
            self.W, self.H = W.T.copy(), H.T.copy()

    def objective(self):
        self.blocking_objectives = getattr(self, "blocking_objectives", 0) + 1
        return orc.klnmf_objective(self.X.T, self.W.T, self.H.T, self.wkl, self.wlh)

    # -- the queued forms (salnmf_objective_async / _read, salnmf_kl_step_keep / _rollback)
    def objective_async(self, slot):
        if not hasattr(self, "ring"):
            self.ring = np.full(256, np.nan)
        self.ring[slot] = orc.klnmf_objective(self.X.T, self.W.T, self.H.T, self.wkl, self.wlh)

    def objective_read(self, first, count):
        self.reads = getattr(self, "reads", [])
        self.reads.append((first, count))
        return self.ring[first : first + count].copy()

    def kl_step_keep(self, n_steps=1, n_given=0):
        self._kept = (self.W.copy(), self.H.copy())
        self.kl_step(n_steps, n_given)
        self.steps_log[-1] = ("keep", n_steps)

    def kl_step_objective(self, slot, n_steps, n_given=0, keep=False):
        self.objective_async(slot)
        self.folded = getattr(self, "folded", 0) + (1 if n_steps > 0 else 0)
        if n_steps == 0:
            return
        if keep:
            self.kl_step_keep(n_steps, n_given)
        else:
            self.kl_step(n_steps, n_given)

    def kl_rollback(self):
        self.W, self.H = self._kept
        self._kept = None
        self.steps_log.append("rollback")

    def update_H(self):
        self.H = orc.update_H(self.X.T, self.W.T, self.H.T, self.wkl, self.wlh).T.copy()

    def samplewise_kl(self):
        return orc.samplewise_kl_divergence(self.X.T, self.W.T, self.H.T)

    def reconstruct(self):
        return self.H @ self.W

    def mv_step(self, n_steps, n_given, lam, delta, gamma):
        self.steps_log.append(n_steps)
        for _ in range(n_steps):
            W, H, gamma = orc.mvnmf_step(self.X.T, self.W.T, self.H.T, lam, delta, gamma, n_given)
            self.W, self.H = W.T.copy(), H.T.copy()
        return gamma

    def mv_step_objective(self, n_steps, n_given, lam, delta, gamma, more_follows=False):
        gamma = self.mv_step(n_steps, n_given, lam, delta, gamma)
        return gamma, self.mv_objective(lam, delta)

    def mv_logdet(self, delta):
        return orc.volume_logdet(self.W.T, delta)

    def mv_update_W_unconstrained(self, n_given, lam, delta):
        return orc.update_W_unconstrained(self.X.T, self.W.T, self.H.T, lam, delta, n_given).T.copy()

    def mv_line_search(self, lam, delta, gamma, W_unconstrained):
        W, H, gamma = orc.line_search(self.X.T, self.W.T, self.H.T, lam, delta, gamma, np.asarray(W_unconstrained).T)
        self.W, self.H = W.T.copy(), H.T.copy()
        return gamma

    def mv_update_W(self, n_given, lam, delta, gamma):
        Wu = orc.update_W_unconstrained(self.X.T, self.W.T, self.H.T, lam, delta, n_given)
        W, H, gamma = orc.line_search(self.X.T, self.W.T, self.H.T, lam, delta, gamma, Wu)
        self.W, self.H = W.T.copy(), H.T.copy()
        return gamma

    def mv_objective(self, lam, delta):
        return orc.kl_divergence_penalized(self.X.T, self.W.T, self.H.T, lam, delta)


    # -- device-side initialisation primitives (salnmf_init_*), restated in NumPy
    def init_gram(self):
        return self.X.T @ self.X, float(self.X.sum())

    def init_project(self, B):
        self.H = self.X @ np.asarray(B).T
        return (np.maximum(self.H, 0) ** 2).sum(axis=0), (np.minimum(self.H, 0) ** 2).sum(axis=0)

    def init_finish(self, scale, take_neg, post, zero_below, fill):
        U = self.H
        E = np.where(np.asarray(take_neg, dtype=bool)[None, :], np.maximum(-U, 0), np.maximum(U, 0)) * np.asarray(scale)[None, :]
        E[:, 0] = np.abs(U[:, 0]) * scale[0]
        E[E < zero_below] = 0
        if fill != 0.0:
            E[E == 0] = fill
        self.H = np.clip(E * np.asarray(post)[None, :], orc.EPSILON, None)

    def init_separable(self, n_select, return_norms=False):
        R = self.X.T / self.X.T.sum(axis=0)
        chosen = np.empty(n_select, dtype=np.int64)
        won = np.empty(n_select)
        for k in range(n_select):
            norms = (R**2).sum(axis=0)
            j = int(np.argmax(norms))
            u = R[:, j]
            R = R - np.outer(u, u @ R) / norms[j]
            chosen[k], won[k] = j, norms[j]
        return (chosen, won) if return_norms else chosen

    def init_flat(self, post):
        e = self.X.sum(axis=1) / self.K
        self.H = np.clip(e[:, None] * np.asarray(post)[None, :], orc.EPSILON, None)


class FakeShardEngine(FakeEngine):
    """Adds the split step (partial / numerator / finish) for the host-collective driver."""

    def kl_step_partial(self):
        import torch

        X, W, H = self.X.T, self.W.T, self.H.T
        aux = X / (W @ H)
        scaled = aux if self.wkl is None else self.wkl * aux
        self._G = torch.from_numpy(np.ascontiguousarray((scaled @ H.T).T))  # (K, V) local numerator
        self.H = orc._h_from_factor(H, W.T @ aux, self.wkl, self.wlh).T.copy()

    def numerator(self):
        return self._G

    def after_collective(self):
        pass

    def kl_step_finish(self, n_given, clip_mode):
        W = self.W.T
        Wn = W * self._G.numpy().T
        Wn = Wn / Wn.sum(axis=0)
        Wn[:, :n_given] = W[:, :n_given]
        self.W = np.clip(Wn, orc.EPSILON, None).T.copy()

    def sync(self):
        pass


class FakeCorrShardEngine(FakeEngine):
    """Oracle-backed CorrNMF engine that speaks the engine's sample-sharding protocol over ``torch.distributed``.

    It restates WHERE the C++ engine exchanges data when a communicator is attached (``salnmf.hip``): an
    all-reduce of the signature-update numerator (``corr_compute_aux``), of the two sums of the signature scalings,
    of the Poisson term and of the sum of squares of the sample embeddings, and ONE gather of ``U``, ``alpha`` and
    ``aux`` per update for the signature-embedding solves, which every rank then runs on identical inputs (the
    engine's regime below 16 384 samples per rank; above it all-reduces the sums of every evaluation round instead).
    Without an initialised process group (or before ``comm_init``) it is a plain single-shard engine.
    """

    def __init__(self, n_samples, n_features, n_signatures, device=0):
        super().__init__(n_samples, n_features, n_signatures, device)
        self.world, self.rank, self.n_total = 1, 0, n_samples
        self._attached = False

    # -- communicator (the unique id is irrelevant here; the process group carries the exchange)
    @staticmethod
    def comm_unique_id():
        return b"\0" * 128

    def p2p_export(self, n_ranks, max_count=None):
        raise RuntimeError("no device memory to export (CPU stand-in)")  # attach_peer_exchange(required=False) -> off

    def comm_init(self, unique_id, n_ranks, rank):
        import torch
        import torch.distributed as dist

        self.world, self.rank, self._attached = n_ranks, rank, True
        t = torch.tensor([self.N], dtype=torch.int64)
        dist.all_reduce(t)
        self.n_total = int(t.item())

    def comm_info(self):
        return self.world, self.rank, self.n_total

    def _allreduce(self, array):
        if not self._attached:
            return np.asarray(array, dtype=float)
        import torch
        import torch.distributed as dist

        t = torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64).copy())
        dist.all_reduce(t)
        return t.numpy()

    def _gather_rows(self, array):
        if not self._attached:
            return np.asarray(array)
        import torch.distributed as dist

        parts = [None] * self.world
        dist.all_gather_object(parts, np.ascontiguousarray(array))
        return np.concatenate(parts, axis=0)

    # -- state
    def corr_configure(self, dim):
        self.dim = dim
        self.alpha, self.beta = np.zeros(self.N), np.zeros(self.K)
        self.L, self.U = np.zeros((self.K, dim)), np.zeros((self.N, dim))
        self.aux = np.zeros((self.N, self.K))

    def corr_upload(self, which, values):
        name = ("beta", "alpha", "L", "U", "aux")[which]
        setattr(self, name, np.array(values, dtype=float))

    def corr_download(self, which):
        return np.array(getattr(self, ("beta", "alpha", "L", "U", "aux")[which]))

    # -- one update, piece by piece (oracle arithmetic on the local shard + the exchanges)
    def corr_update_sample_scalings(self):
        from oracle import corrnmf_oracle as corr

        self.alpha = corr.update_sample_scalings(self.X, self.beta, self.L, self.U)

    def corr_compute_exposures(self):
        from oracle import corrnmf_oracle as corr

        self.H = corr.compute_exposures(self.beta, self.alpha, self.L, self.U)  # (N, K)

    def corr_compute_aux(self):
        from oracle import corrnmf_oracle as corr

        aux_kn = corr.compute_aux(self.X, self.W, self.H)  # (K, N)
        self.aux = aux_kn.T.copy()
        ratio = self.X / (self.H @ self.W)  # (N, V)
        self._G = self._allreduce(self.H.T @ ratio)  # (K, V) numerator of update_signatures, all shards

    def corr_update_signature_scalings(self):
        first = self._allreduce(self.aux.sum(axis=0))
        second = self._allreduce(np.exp(self.alpha[None, :] + self.L @ self.U.T).sum(axis=1))
        self.beta = np.log(first) - np.log(second)

    def corr_update_signature_embeddings(self, variance, maxiter=0, return_status=False):
        from oracle import corrnmf_oracle as corr

        U_all, alpha_all, aux_all = self._gather_rows(self.U), self._gather_rows(self.alpha), self._gather_rows(self.aux)
        self.L = corr.update_signature_embeddings(aux_all.T, self.L, U_all, self.beta, alpha_all, variance)

    def corr_update_sample_embeddings(self, variance, maxiter=3, return_status=False):
        from oracle import corrnmf_oracle as corr

        self.U = corr.update_sample_embeddings(self.aux.T, self.L, self.U, self.beta, self.alpha, variance)

    @staticmethod
    def corr_update_sample_embeddings_multi(engines, variance, maxiter=3, return_status=False):
        from oracle import corrnmf_oracle as corr

        U = corr.mm_update_sample_embeddings(
            [e.aux.T for e in engines], [e.L for e in engines], engines[0].U, [e.beta for e in engines], [e.alpha for e in engines], variance
        )
        for e in engines:
            e.U = U.copy()

    def corr_embedding_sumsq(self):
        return float(np.sum(self.L**2)), float(self._allreduce(np.array([np.sum(self.U**2)]))[0])

    def corr_update_signatures(self, n_given=0):
        Wn = self.W * self._G
        Wn = Wn / Wn.sum(axis=1, keepdims=True)
        Wn[:n_given] = self.W[:n_given]
        Wn[n_given:] = np.clip(Wn[n_given:], orc.EPSILON, None)  # update_W clips the non-given rows only
        self.W = Wn

    def corr_poisson_llh(self):
        from oracle import corrnmf_oracle as corr

        return float(self._allreduce(np.array([corr.poisson_llh(self.X.T, self.W.T, self.H.T)]))[0])
