"""Argument checkers and small host helpers with the reference's error behaviour.

Mirrors ``src/salamander/utils.py:16-99`` (checker semantics and exception types) and
``utils.py:155-158`` (``normalize_WH``).  Only what the KLNMF / MvNMF fit path touches.
"""

from __future__ import annotations

from typing import Any, Iterable

import numpy as np
import pandas as pd

EPSILON = np.finfo(np.float32).eps  # src/salamander/utils.py:13


def type_checker(arg_name: str, arg: Any, allowed_types) -> None:
    """Exact-type test (``type(arg) in allowed``), ``TypeError`` otherwise -- utils.py:61-77."""
    allowed = [allowed_types] if isinstance(allowed_types, type) else list(allowed_types)
    if type(arg) not in allowed:
        raise TypeError(f"The type of '{arg_name}' has to be one of {allowed}.")


def shape_checker(arg_name: str, arg, allowed_shape: tuple[int, ...]) -> None:
    """ndarray / DataFrame of exactly ``allowed_shape``, ``ValueError`` otherwise -- utils.py:39-58."""
    type_checker(arg_name, arg, [np.ndarray, pd.DataFrame])
    if tuple(arg.shape) != tuple(allowed_shape):
        raise ValueError(f"The shape of '{arg_name}' has to be {allowed_shape}.")


def value_checker(arg_name: str, arg: Any, allowed_values: Iterable[Any]) -> None:
    """Membership test, ``ValueError`` otherwise -- utils.py:80-99."""
    if isinstance(allowed_values, type):
        allowed_values = [allowed_values]
    if arg not in allowed_values:
        raise ValueError(f"The value of '{arg_name}' has to be one of {allowed_values}.")


def dict_checker(dict_name: str, dictionary: dict, valid_keys: list) -> None:
    """A dict whose keys all lie in ``valid_keys`` -- utils.py:16-36."""
    type_checker(dict_name, dictionary, dict)
    extra = [k for k in dictionary if k not in valid_keys]
    if extra:
        raise ValueError(f"'{dict_name}' includes keys outside of {valid_keys}.")


def normalize_WH(W: np.ndarray, H: np.ndarray):
    """Unit column sums for ``W (V, K)``, compensated in ``H (K, N)`` -- utils.py:155-158.

    One-time host step at initialisation; inside the MvNMF loop the same operation runs on
    the device (``mv_trial_kernel`` / ``scale_H_kernel``).
    """
    s = W.sum(axis=0)
    return W / s, H * s[:, None]
