"""MI355X-native KL-NMF update engine behind Salamander's ``KLNMF`` / ``MvNMF`` API.

``import salamander_amd as sal; sal.models.KLNMF(n_signatures=50).fit(adata)`` is a
drop-in for the reference's fit path.  Importing the package needs
no GPU; every compute entry point raises :class:`EngineUnavailable` when the HIP
extension or a gfx950 device is missing -- there is no CPU fallback.
"""

from . import models
from ._lib import EngineUnavailable
from .anndata_compat import AnnData, MuData
from .engine import Engine

__version__ = "0.1.0"
__all__ = ["models", "Engine", "AnnData", "MuData", "EngineUnavailable"]
