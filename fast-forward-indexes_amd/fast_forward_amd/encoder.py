"""Query encoders: the injection point for query vectors.

Only the abstract interface and the function adapter of the reference are provided
(``src/fast_forward/encoder/base.py:10-23``, ``encoder/__init__.py:32-44``).  The HuggingFace
transformer encoders are model inference and out of scope for this package (SURVEY.md section 2,
row 7): wrap any model in ``LambdaEncoder`` or subclass ``Encoder``.
"""

from __future__ import annotations

import abc
from collections.abc import Callable, Sequence

import numpy as np


class Encoder(abc.ABC):
    """Maps a batch of texts to a ``(len(texts), dim)`` array."""

    @abc.abstractmethod
    def _encode(self, texts: Sequence[str]) -> np.ndarray:
        ...

    def __call__(self, texts: Sequence[str]) -> np.ndarray:
        return self._encode(texts)


class LambdaEncoder(Encoder):
    """Adapter for a function that encodes one text."""

    def __init__(self, f: Callable[[str], np.ndarray]) -> None:
        super().__init__()
        self._f = f

    def _encode(self, texts: Sequence[str]) -> np.ndarray:
        return np.array([self._f(t) for t in texts])


class ArrayEncoder(Encoder):
    """Pre-computed query vectors keyed by query text (a ``dict`` or a callable batch function).

    Not in the reference; the cheapest way to feed an externally encoded ``(Q, d)`` matrix."""

    def __init__(self, vectors_by_text: dict[str, np.ndarray]) -> None:
        super().__init__()
        self._table = vectors_by_text

    def _encode(self, texts: Sequence[str]) -> np.ndarray:
        return np.stack([self._table[t] for t in texts])
