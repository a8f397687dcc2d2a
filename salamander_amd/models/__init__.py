"""Model classes with the reference's names (``src/salamander/models/__init__.py:5-15``).

The KL hot path: ``KLNMF`` and ``MvNMF``; SURVEY.md section 8 row f1: ``CorrNMFDet`` (dense pieces
on the device, embedding solves via SciPy as in the reference).  ``MultimodalCorrNMF`` is not built.
"""

from . import _utils_corrnmf, _utils_klnmf, corrnmf_det
from .corrnmf_det import CorrNMFDet
from .klnmf import KLNMF
from .mvnmf import MvNMF

__all__ = ["KLNMF", "MvNMF", "CorrNMFDet", "corrnmf_det", "_utils_klnmf", "_utils_corrnmf"]
