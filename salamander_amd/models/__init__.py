"""Model classes with the reference's names (``src/salamander/models/__init__.py:5-15``).

The KL hot path: ``KLNMF`` and ``MvNMF``; SURVEY.md section 8 row f1: ``CorrNMFDet`` and
``MultimodalCorrNMF`` (dense pieces and the Newton-CG embedding solves on the device).
"""

from . import _utils_corrnmf, _utils_klnmf, corrnmf_det, mmcorrnmf
from .corrnmf_det import CorrNMFDet
from .klnmf import KLNMF
from .mmcorrnmf import MultimodalCorrNMF
from .mvnmf import MvNMF

__all__ = ["KLNMF", "MvNMF", "CorrNMFDet", "MultimodalCorrNMF", "corrnmf_det", "mmcorrnmf", "_utils_klnmf", "_utils_corrnmf"]
