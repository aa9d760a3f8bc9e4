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
    """Pre-computed query vectors keyed by query text (a ``dict``).

    Not in the reference; the cheapest way to feed an externally encoded ``(Q, d)`` matrix.  The vectors are laid out as
    one matrix on first use, and a batch is one gather of its rows; ``Index.encode_queries`` hands it all texts at once
    (``encode_all``: there is no model behind it whose batch size would matter) -- 6980 queries: ~1 ms instead of 5 - 8
    for 219 batches of 32."""

    def __init__(self, vectors_by_text: dict[str, np.ndarray]) -> None:
        super().__init__()
        self._table = vectors_by_text
        self._rows: dict[str, int] | None = None
        self._matrix: np.ndarray | None = None

    def _laid_out(self) -> None:
        if self._rows is None or len(self._rows) != len(self._table):
            self._rows = {t: i for i, t in enumerate(self._table)}
            self._matrix = np.stack(list(self._table.values())) if self._table else np.zeros((0, 0), dtype=np.float32)

    def _encode(self, texts: Sequence[str]) -> np.ndarray:
        self._laid_out()
        rows = self._rows
        return self._matrix[np.fromiter((rows[t] for t in texts), dtype=np.int64, count=len(texts))]

    encode_all = _encode
