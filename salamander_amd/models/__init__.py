"""Model classes with the reference's names (``src/salamander/models/__init__.py:5-15``).

Only the KL hot path is implemented: ``KLNMF`` and ``MvNMF``.  ``CorrNMFDet`` and
``MultimodalCorrNMF`` are out of scope.
"""

from . import _utils_klnmf
from .klnmf import KLNMF
from .mvnmf import MvNMF

__all__ = ["KLNMF", "MvNMF", "_utils_klnmf"]
