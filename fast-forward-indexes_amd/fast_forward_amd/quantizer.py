"""Quantizers: the ``Quantizer`` protocol of the reference and product quantizers behind it.

API as in ``fast_forward.quantizer`` (reference ``src/fast_forward/quantizer/base.py:16-196``,
``quantizer/nanopq.py:9-149``): ``fit`` / ``encode`` / ``decode``, ``dims``, ``dtype``, the
attach-once rule, ``serialize`` / ``deserialize``.

The reference delegates the arithmetic to the third-party package **nanopq 0.2.1**, which is not
vendored and not installed here, so ``NanoPQ`` / ``NanoOPQ`` below re-implement nanopq's
published algorithm on scipy (per-subspace ``kmeans2(minit="points")`` with seed 123, nearest
codeword by L2, decode = concatenated codewords; OPQ = alternating PQ fit and an SVD rotation
update).  Parity of ``fit`` / ``encode`` with nanopq is therefore **unpinned**; what the index does
with a fitted quantizer -- ``decode`` then score, ``index/base.py:292-293`` -- is pinned against the
reference itself (tests/golden/pq_*.npz, produced by handing the reference a quantizer with the
same codewords).
"""

from __future__ import annotations

import abc
import importlib
import logging
from collections.abc import Mapping
from typing import Any

import numpy as np

LOGGER = logging.getLogger(__name__)

QuantizerAttributes = Mapping[str, "str | bool | float"]
QuantizerData = Mapping[str, np.ndarray]


class Quantizer(abc.ABC):
    """Base class of quantizers (vector -> code -> approximate vector)."""

    _attached: bool = False
    _trained: bool = False

    # ---- to implement -------------------------------------------------------------------------
    @abc.abstractmethod
    def _fit(self, vectors: np.ndarray, **kwargs: Any) -> None: ...

    @abc.abstractmethod
    def _get_dtype(self) -> np.dtype: ...

    @abc.abstractmethod
    def _get_dims(self) -> tuple[int | None, int | None]: ...

    @abc.abstractmethod
    def _encode(self, vectors: np.ndarray) -> np.ndarray: ...

    @abc.abstractmethod
    def _decode(self, codes: np.ndarray) -> np.ndarray: ...

    @abc.abstractmethod
    def _get_state(self) -> tuple[QuantizerAttributes, QuantizerData]: ...

    @classmethod
    @abc.abstractmethod
    def _from_state(cls, attributes: QuantizerAttributes, data: QuantizerData) -> "Quantizer": ...

    # ---- protocol -------------------------------------------------------------------------------
    def set_attached(self) -> None:
        """Called when an index takes the quantizer; an untrained quantizer cannot be attached."""
        if not self._trained:
            raise RuntimeError(f"Call {self.__class__.__name__}.fit before attaching the quantizer to an index.")
        self._attached = True

    def fit(self, vectors: np.ndarray, **kwargs: Any) -> None:
        """Train; refused once attached to an index (its codes would go stale)."""
        if self._attached:
            raise RuntimeError("Quantizers can only be fitted before they are attached to an index.")
        self._fit(vectors, **kwargs)
        self._trained = True

    def encode(self, vectors: np.ndarray) -> np.ndarray:
        if not self._trained:
            raise RuntimeError(f"Call {self.__class__.__name__}.fit first.")
        return self._encode(vectors)

    def decode(self, codes: np.ndarray) -> np.ndarray:
        if not self._trained:
            raise RuntimeError(f"Call {self.__class__.__name__}.fit first.")
        return self._decode(codes)

    @property
    def dtype(self) -> np.dtype:
        """dtype of the codes."""
        return self._get_dtype()

    @property
    def dims(self) -> tuple[int | None, int | None]:
        """(dimension of the vectors, length of the codes); entries are None before training."""
        return self._get_dims()

    def serialize(self) -> tuple[QuantizerAttributes, QuantizerAttributes, QuantizerData]:
        """(meta, attributes, data): what an index stores to rebuild the quantizer."""
        meta = {"__module__": self.__class__.__module__, "__name__": self.__class__.__name__, "_trained": self._trained}
        attributes, data = self._get_state()
        return meta, attributes, data

    @classmethod
    def deserialize(cls, meta: QuantizerAttributes, attributes: QuantizerAttributes, data: QuantizerData) -> "Quantizer":
        LOGGER.debug("reconstructing %s.%s", meta["__module__"], meta["__name__"])
        q_cls = getattr(importlib.import_module(str(meta["__module__"])), str(meta["__name__"]))
        quantizer = q_cls._from_state(attributes, data)
        quantizer._trained = bool(meta["_trained"])
        return quantizer

    def __eq__(self, o: object) -> bool:
        if not isinstance(o, Quantizer):
            return False
        mine, theirs = self.serialize(), o.serialize()
        if mine[0] != theirs[0] or mine[1] != theirs[1] or mine[2].keys() != theirs[2].keys():
            return False
        return all(np.array_equal(v, theirs[2][k]) for k, v in mine[2].items())

    __hash__ = None

    # ---- what the device path needs (not in the reference) -----------------------------------------
    def device_codebook(self) -> tuple[np.ndarray, np.ndarray | None] | None:
        """``(codewords float32 [M, Ks, Ds], rotation float32 [d, d] or None)`` if this quantizer
        is a product quantizer the HIP kernels can score directly; ``None`` otherwise."""
        return None


def _code_dtype(ks: int) -> np.dtype:
    return np.dtype(np.uint8 if ks <= 2**8 else (np.uint16 if ks <= 2**16 else np.uint32))


class _PQCore:
    """Product quantization: M subspaces of Ds = d/M dimensions, Ks codewords each."""

    def __init__(self, M: int, Ks: int, metric: str = "dot", verbose: bool = False) -> None:
        assert 0 < Ks <= 2**32
        assert metric in ("l2", "dot")
        self.M, self.Ks, self.metric, self.verbose = int(M), int(Ks), metric, bool(verbose)
        self.code_dtype = _code_dtype(self.Ks)
        self.codewords: np.ndarray | None = None
        self.Ds: int | None = None

    def fit(self, vecs: np.ndarray, iter: int = 20, seed: int = 123, minit: str = "points") -> "_PQCore":
        from scipy.cluster.vq import kmeans2

        assert vecs.dtype == np.float32 and vecs.ndim == 2
        n, d = vecs.shape
        assert self.Ks < n, "the number of training vectors should be more than Ks"
        assert d % self.M == 0, "input dimension must be dividable by M"
        self.Ds = d // self.M
        np.random.seed(seed)
        self.codewords = np.zeros((self.M, self.Ks, self.Ds), dtype=np.float32)
        for m in range(self.M):
            if self.verbose:
                LOGGER.info("training the subspace: %s / %s", m, self.M)
            sub = vecs[:, m * self.Ds:(m + 1) * self.Ds]
            init = self.codewords[m] if minit == "matrix" else self.Ks
            self.codewords[m], _ = kmeans2(sub, init, iter=iter, minit=minit)
        return self

    def encode(self, vecs: np.ndarray) -> np.ndarray:
        from scipy.cluster.vq import vq

        assert vecs.dtype == np.float32 and vecs.ndim == 2
        n, d = vecs.shape
        assert d == self.Ds * self.M, "input dimension must be Ds * M"
        codes = np.empty((n, self.M), dtype=self.code_dtype)
        for m in range(self.M):
            codes[:, m], _ = vq(vecs[:, m * self.Ds:(m + 1) * self.Ds], self.codewords[m])
        return codes

    def decode(self, codes: np.ndarray) -> np.ndarray:
        assert codes.ndim == 2 and codes.shape[1] == self.M and codes.dtype == self.code_dtype
        out = np.empty((codes.shape[0], self.Ds * self.M), dtype=np.float32)
        for m in range(self.M):
            out[:, m * self.Ds:(m + 1) * self.Ds] = self.codewords[m][codes[:, m], :]
        return out


class NanoPQ(Quantizer):
    """Product quantizer with the interface of the reference's ``NanoPQ`` (``nanopq.PQ``)."""

    def __init__(self, M: int, Ks: int, metric: str = "dot", verbose: bool = False) -> None:
        self._pq = _PQCore(M=M, Ks=Ks, metric=metric, verbose=verbose)
        super().__init__()

    def _fit(self, vectors: np.ndarray, **kwargs: Any) -> None:
        self._pq.fit(vectors, **kwargs)

    def _get_dtype(self) -> np.dtype:
        return np.dtype(self._pq.code_dtype)

    def _get_dims(self) -> tuple[int | None, int | None]:
        return (None if self._pq.Ds is None else self._pq.Ds * self._pq.M), self._pq.M

    def _encode(self, vectors: np.ndarray) -> np.ndarray:
        return self._pq.encode(vectors)

    def _decode(self, codes: np.ndarray) -> np.ndarray:
        return self._pq.decode(codes)

    def _get_state(self) -> tuple[QuantizerAttributes, QuantizerData]:
        attributes = {a: getattr(self._pq, a) for a in ("M", "Ks", "Ds", "metric", "verbose")}
        data = {} if self._pq.codewords is None else {"codewords": self._pq.codewords}
        return attributes, data

    @classmethod
    def _from_state(cls, attributes: QuantizerAttributes, data: QuantizerData) -> "NanoPQ":
        q = cls(M=int(attributes["M"]), Ks=int(attributes["Ks"]), metric=str(attributes["metric"]), verbose=bool(attributes["verbose"]))
        if attributes.get("Ds") is not None:
            q._pq.Ds = int(attributes["Ds"])
        if "codewords" in data:
            q._pq.codewords = np.asarray(data["codewords"], dtype=np.float32)
        return q

    @classmethod
    def from_codewords(cls, codewords: np.ndarray, metric: str = "dot") -> "NanoPQ":
        """A trained quantizer around existing codewords ``[M, Ks, Ds]`` (e.g. loaded from an index)."""
        codewords = np.ascontiguousarray(codewords, dtype=np.float32)
        m, ks, ds = codewords.shape
        q = cls(M=m, Ks=ks, metric=metric)
        q._pq.Ds, q._pq.codewords, q._trained = ds, codewords, True
        return q

    def device_codebook(self):
        return (self._pq.codewords, None) if self._trained and self._pq.code_dtype == np.uint8 else None


class NanoOPQ(Quantizer):
    """Optimized product quantizer (a learned rotation ``R`` in front of a product quantizer),
    interface of the reference's ``NanoOPQ`` (``nanopq.OPQ``): encode ``x R``, decode ``c R^T``."""

    def __init__(self, M: int, Ks: int, metric: str = "dot", verbose: bool = False) -> None:
        self._pq = _PQCore(M=M, Ks=Ks, metric=metric, verbose=verbose)
        self._R: np.ndarray | None = None
        super().__init__()

    def _fit(self, vectors: np.ndarray, pq_iter: int = 20, rotation_iter: int = 10, seed: int = 123, **_: Any) -> None:
        assert vectors.dtype == np.float32 and vectors.ndim == 2
        d = vectors.shape[1]
        rot = np.eye(d, dtype=np.float32)
        for i in range(rotation_iter):
            x = vectors @ rot
            last = i == rotation_iter - 1
            tmp = _PQCore(M=self._pq.M, Ks=self._pq.Ks, metric=self._pq.metric, verbose=self._pq.verbose)
            if i == 0:
                tmp.fit(x, iter=pq_iter if last else 1, seed=seed, minit="points")
            else:  # warm start from the previous codewords
                tmp.Ds, tmp.codewords = self._pq.Ds, self._pq.codewords.copy()
                tmp.fit(x, iter=pq_iter if last else 1, seed=seed, minit="matrix")
            self._pq.Ds, self._pq.codewords = tmp.Ds, tmp.codewords
            if last:
                break
            approx = tmp.decode(tmp.encode(x))
            u, _, vt = np.linalg.svd(vectors.T @ approx)
            rot = (u @ vt).astype(np.float32)
        self._R = rot

    def _get_dtype(self) -> np.dtype:
        return np.dtype(self._pq.code_dtype)

    def _get_dims(self) -> tuple[int | None, int | None]:
        return (None if self._pq.Ds is None else self._pq.Ds * self._pq.M), self._pq.M

    def _encode(self, vectors: np.ndarray) -> np.ndarray:
        return self._pq.encode((vectors @ self._R).astype(np.float32))

    def _decode(self, codes: np.ndarray) -> np.ndarray:
        return self._pq.decode(codes) @ self._R.T

    def _get_state(self) -> tuple[QuantizerAttributes, QuantizerData]:
        attributes = {a: getattr(self._pq, a) for a in ("M", "Ks", "Ds", "metric", "verbose")}
        data = {}
        if self._pq.codewords is not None:
            data["codewords"] = self._pq.codewords
        if self._R is not None:
            data["R"] = self._R
        return attributes, data

    @classmethod
    def _from_state(cls, attributes: QuantizerAttributes, data: QuantizerData) -> "NanoOPQ":
        q = cls(M=int(attributes["M"]), Ks=int(attributes["Ks"]), metric=str(attributes["metric"]), verbose=bool(attributes["verbose"]))
        if attributes.get("Ds") is not None:
            q._pq.Ds = int(attributes["Ds"])
        if "codewords" in data:
            q._pq.codewords = np.asarray(data["codewords"], dtype=np.float32)
        if "R" in data:
            q._R = np.asarray(data["R"], dtype=np.float32)
        return q

    def device_codebook(self):
        if not self._trained or self._pq.code_dtype != np.uint8:
            return None
        return self._pq.codewords, self._R
