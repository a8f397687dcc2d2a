"""``StandardNMF``: models parameterised by a signature and an exposure matrix.

Mirrors ``src/salamander/models/standard_nmf.py:32-58`` (``_initialize``).  The embedding
plots of the reference class are out of scope.
"""

from __future__ import annotations

from typing import Any

from ..initialization import initialize_standard_nmf
from .signature_nmf import SignatureNMF


class StandardNMF(SignatureNMF):
    def _initialize(self, given_parameters: dict[str, Any] | None = None, init_kwargs: dict[str, Any] | None = None) -> None:
        """Initialise signatures and exposures; given signatures are never overwritten later."""
        init_kwargs = {} if init_kwargs is None else init_kwargs.copy()
        self.asignatures = initialize_standard_nmf(
            self.adata, self.n_signatures, self.init_method, given_parameters, **init_kwargs
        )
