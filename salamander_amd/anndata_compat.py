"""AnnData in / AnnData out, with or without the ``anndata`` package.

The reference stores everything in ``anndata.AnnData`` (``signature_nmf.py:269-281``,
``initialize.py:206-216``).  ``anndata`` is an optional dependency here: when it is
importable the real class is used; otherwise :class:`MiniAnnData` supplies exactly the
attribute surface the fit path touches: ``X`` get/set, ``obs``,
``obsm``, ``obsp``, ``obs_names``, ``var_names``, ``n_obs``, ``n_vars``, ``to_df()``, ``copy()``
and row slicing ``adata[:n, :]``.
"""

from __future__ import annotations

import numpy as np
import pandas as pd

try:  # pragma: no cover - depends on the environment
    import anndata as _ad

    _REAL = _ad.AnnData
except Exception:  # anndata is absent in the build container and may be absent on the GPU box
    _ad = None
    _REAL = None


class MiniAnnData:
    """Minimal annotated matrix: ``X (n_obs, n_vars)`` + names + ``obs`` / ``obsm``."""

    def __init__(self, X=None, obs_names=None, var_names=None):
        if isinstance(X, pd.DataFrame):
            obs_names = X.index if obs_names is None else obs_names
            var_names = X.columns if var_names is None else var_names
            X = X.values
        if X is None:
            X = np.zeros((0, 0))
        X = np.asarray(X)
        if X.ndim != 2:
            raise ValueError("X has to be two-dimensional.")
        self._X = X
        n_obs, n_vars = X.shape
        self.obs = pd.DataFrame(index=self._names(obs_names, n_obs))
        self._var_names = self._names(var_names, n_vars)
        self.obsm: dict = {}
        self.obsp: dict = {}

    @staticmethod
    def _names(names, n):
        if names is None:
            return pd.Index([str(i) for i in range(n)])
        names = pd.Index(names)
        if len(names) != n:
            raise ValueError("Length of names does not match the matrix.")
        return names.astype(str)

    @property
    def X(self):
        return self._X

    @X.setter
    def X(self, value):
        value = np.asarray(value)
        if value.shape != self._X.shape:
            raise ValueError("Shape of X must not change.")
        self._X = value

    @property
    def n_obs(self):
        return self._X.shape[0]

    @property
    def n_vars(self):
        return self._X.shape[1]

    @property
    def shape(self):
        return self._X.shape

    @property
    def obs_names(self):
        return self.obs.index

    @obs_names.setter
    def obs_names(self, names):
        self.obs.index = self._names(names, self.n_obs)

    @property
    def var_names(self):
        return self._var_names

    @var_names.setter
    def var_names(self, names):
        self._var_names = self._names(names, self.n_vars)

    def to_df(self) -> pd.DataFrame:
        return pd.DataFrame(self._X, index=self.obs_names, columns=self.var_names)

    def copy(self):
        new = MiniAnnData(self._X.copy(), self.obs_names.copy(), self.var_names.copy())
        new.obs = self.obs.copy()
        new.obsm = {k: np.array(v, copy=True) for k, v in self.obsm.items()}
        return new

    def __getitem__(self, index):
        rows, cols = index if isinstance(index, tuple) else (index, slice(None))
        ridx = np.arange(self.n_obs)[rows]
        cidx = np.arange(self.n_vars)[cols]
        ridx, cidx = np.atleast_1d(ridx), np.atleast_1d(cidx)
        new = MiniAnnData(self._X[np.ix_(ridx, cidx)], self.obs_names[ridx], self.var_names[cidx])
        new.obs = self.obs.iloc[ridx].copy()
        new.obsm = {k: np.asarray(v)[ridx] for k, v in self.obsm.items()}
        return new

    def __repr__(self):
        return f"MiniAnnData object with n_obs x n_vars = {self.n_obs} x {self.n_vars}"


class MiniMuData:
    """Minimal multimodal container: named modalities over the same samples + shared ``obsm``.

    The attribute surface ``MultimodalCorrNMF`` touches (``mmcorrnmf.py:61-64,200-215,490``):
    ``mod`` (dict name -> AnnData), ``mdata[name]``, ``obsm``, ``obs_names``, ``n_obs``, ``n_mod``, ``update()``.
    """

    def __init__(self, mods: dict):
        self.mod = dict(mods)
        self.obsm: dict = {}

    def __getitem__(self, name):
        return self.mod[name]

    @property
    def n_mod(self) -> int:
        return len(self.mod)

    @property
    def obs_names(self):
        first = next(iter(self.mod.values()), None)
        return pd.Index([]) if first is None else first.obs_names

    @property
    def n_obs(self) -> int:
        return len(self.obs_names)

    def update(self) -> None:
        """The real MuData refreshes its global annotations here; nothing is cached in this container."""

    def __repr__(self):
        return f"MiniMuData object with n_obs = {self.n_obs} and modalities {list(self.mod)}"


try:  # pragma: no cover - depends on the environment
    import mudata as _md

    _REAL_MU = _md.MuData
except Exception:
    _md = None
    _REAL_MU = None

# The class new containers are built with, and the classes accepted as input.
AnnData = _REAL if _REAL is not None else MiniAnnData
ANNDATA_TYPES = [c for c in (_REAL, MiniAnnData) if c is not None]
MuData = _REAL_MU if _REAL_MU is not None else MiniMuData
MUDATA_TYPES = [c for c in (_REAL_MU, MiniMuData) if c is not None]


def concat_rows(first, second):
    """Row-wise concatenation keeping names (``ad.concat([...], join='outer')``, initialize.py:214-216)."""
    if _REAL is not None and isinstance(first, _REAL) and isinstance(second, _REAL):
        return _ad.concat([first, second], join="outer")
    if list(first.var_names) != list(second.var_names):
        raise ValueError("concat_rows needs identical var_names.")
    out = MiniAnnData(
        np.concatenate([np.asarray(first.X, dtype=float), np.asarray(second.X, dtype=float)], axis=0),
        list(first.obs_names) + list(second.obs_names),
        first.var_names,
    )
    return out
