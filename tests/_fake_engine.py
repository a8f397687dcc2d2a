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

    def set_weights(self, wkl=None, wlh=None):
        self.wkl, self.wlh = wkl, wlh

    def download_W(self):
        return self.W.copy()

    def download_H(self):
        return self.H.copy()

    def kl_step(self, n_steps=1, n_given=0):
        self.steps_log.append(n_steps)
        for _ in range(n_steps):
            W, H = orc.update_WH(self.X.T, self.W.T, self.H.T, self.wkl, self.wlh, n_given)
            self.W, self.H = W.T.copy(), H.T.copy()

    def objective(self):
        return orc.klnmf_objective(self.X.T, self.W.T, self.H.T, self.wkl, self.wlh)

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

    def mv_update_W(self, n_given, lam, delta, gamma):
        Wu = orc.update_W_unconstrained(self.X.T, self.W.T, self.H.T, lam, delta, n_given)
        W, H, gamma = orc.line_search(self.X.T, self.W.T, self.H.T, lam, delta, gamma, Wu)
        self.W, self.H = W.T.copy(), H.T.copy()
        return gamma

    def mv_objective(self, lam, delta):
        return orc.kl_divergence_penalized(self.X.T, self.W.T, self.H.T, lam, delta)


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
