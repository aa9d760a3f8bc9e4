"""Helpers on the ``Index.add`` side of the path (the reference's ``fast_forward.util``):

* :class:`Indexer` -- feed an index from text dictionaries or from another index, optionally
  fitting a quantizer on the first batches (util/indexer.py:28-178);
* :func:`create_coalesced_index` -- sequential coalescing of a document's passage vectors
  (util/__init__.py:51-101);
* :func:`to_ir_measures`, :func:`cos_dist` (util/__init__.py:29-48).

Host-side only: everything here ends in ``Index.add``.
"""

from __future__ import annotations

import logging
from collections.abc import Callable, Iterable, Sequence
from typing import TypedDict

import numpy as np

LOGGER = logging.getLogger(__name__)

__all__ = ["Indexer", "IndexingDict", "to_ir_measures", "cos_dist", "create_coalesced_index"]


class IndexingDict(TypedDict, total=False):
    """One document/passage for :meth:`Indexer.from_dicts`; ``doc_id`` / ``psg_id`` may be absent."""

    text: str
    doc_id: "str | None"
    psg_id: "str | None"


def to_ir_measures(ranking):
    """The ranking as the frame the ir-measures library expects (query_id, doc_id, score)."""
    return ranking._df[["q_id", "id", "score"]].rename(columns={"q_id": "query_id", "id": "doc_id"})


def cos_dist(a: np.ndarray, b: np.ndarray) -> float:
    """Cosine distance of two 1-D vectors."""
    if a.ndim != 1 or b.ndim != 1:
        raise AssertionError("cos_dist takes two 1-D vectors")
    return float(1 - np.dot(a, b) / (np.linalg.norm(a) * np.linalg.norm(b)))


def _coalesce(passages: np.ndarray, delta: float, distance: Callable[[np.ndarray, np.ndarray], float]) -> list[np.ndarray]:
    """Sequential coalescing of one document: walk the passages in order, keep a running group
    whose representative is the mean of its members, close the group when the next passage is at
    least ``delta`` away from that mean.  The mean is a running sum in the vectors' dtype divided
    by the count -- the same arithmetic as ``np.mean(group, axis=0)`` (row-by-row accumulation)."""
    out: list[np.ndarray] = []
    acc = np.array(passages[0], copy=True)
    count = 1
    mean = acc / count
    for v in passages[1:]:
        if distance(v, mean) >= delta:
            out.append(mean)
            acc = np.array(v, copy=True)
            count = 1
        else:
            acc = acc + v
            count += 1
        mean = acc / count
    out.append(mean)
    return out


def create_coalesced_index(source_index, target_index, delta: float,
                           distance_function: Callable[[np.ndarray, np.ndarray], float] = cos_dist,
                           batch_size: int | None = None) -> None:
    """Fill the empty ``target_index`` with the coalesced passage vectors of every document of
    ``source_index`` (which are looked up under the source's current mode, as in the reference).

    :param batch_size: vectors to collect before each ``target_index.add`` (default: one add at the end).
    :raises ValueError: when the target index is not empty.
    """
    if len(target_index) > 0:
        raise ValueError("Target index is not empty.")
    doc_ids = _ordered_doc_ids(source_index)
    limit = batch_size or len(doc_ids)
    vectors: list[np.ndarray] = []
    owners: list[str] = []

    def flush() -> None:
        if vectors:
            target_index.add(np.array(vectors), doc_ids=list(owners))
            vectors.clear()
            owners.clear()

    # the source rows live in HBM (or on disk): fetch the passages of many documents per look-up
    step = 4096
    for lo in range(0, len(doc_ids), step):
        rows, row_owner = source_index._get_vectors(doc_ids[lo:lo + step])
        row_owner = np.asarray(row_owner, dtype=object)
        start = 0
        while start < len(row_owner):
            end = start + 1
            while end < len(row_owner) and row_owner[end] == row_owner[start]:
                end += 1
            for rep in _coalesce(rows[start:end], delta, distance_function):
                if len(vectors) == limit:
                    flush()
                vectors.append(rep)
                owners.append(row_owner[start])
            start = end
    flush()
    assert source_index.doc_ids == target_index.doc_ids


def _ordered_doc_ids(index) -> list[str]:
    """Document ids in insertion order when the index can tell, otherwise sorted (the reference
    walks a set, i.e. in no particular order)."""
    seen: dict[str, None] = {}
    try:
        for _, docs, _ in index._batch_iter(2**16):
            for d in docs:
                if d is not None:
                    seen.setdefault(d, None)
        return list(seen)
    except (AttributeError, NotImplementedError):
        return sorted(index.doc_ids)


class Indexer:
    """Adds batches of vectors to an index; with ``quantizer`` given, buffers the first
    ``quantizer_fit_batches`` batches, fits the quantizer on them, attaches it to the (still
    empty) index and only then adds what was buffered."""

    def __init__(self, index, encoder=None, encoder_batch_size: int = 128, batch_size: int = 2**16, quantizer=None,
                 quantizer_fit_batches: int = 1) -> None:
        """
        :raises ValueError: when the quantizer has been fit already, or the index is not empty
            while a quantizer is to be attached.
        """
        self._index, self._encoder = index, encoder
        self._encoder_batch_size, self._batch_size = encoder_batch_size, batch_size
        self._pending_quantizer, self._fit_batches = quantizer, quantizer_fit_batches
        self._held: list[tuple[np.ndarray, Sequence | None, Sequence | None]] = []
        if quantizer is None:
            return
        if quantizer._trained:
            raise ValueError("The quantizer is already fit. It should be attached to the index directly.")
        if len(index) > 0:
            raise ValueError("The index must be empty for a quantizer to be attached.")
        if quantizer_fit_batches > 1:
            LOGGER.warning("inputs will be buffered and index will remain empty until the quantizer has been fit")

    def _index_batch(self, vectors: np.ndarray, doc_ids=None, psg_ids=None) -> None:
        if self._pending_quantizer is None:
            self._index.add(vectors, doc_ids, psg_ids)
            return
        self._held.append((vectors, doc_ids, psg_ids))
        if len(self._held) < self._fit_batches:
            return
        LOGGER.info("fitting quantizer (%s batch(es), batch size %s)", len(self._held), self._batch_size)
        if self._held[-1][0].shape[0] < self._batch_size:
            LOGGER.warning("the size of the last batch (%s) is smaller than %s", self._held[-1][0].shape[0], self._batch_size)
        quantizer, self._pending_quantizer = self._pending_quantizer, None
        quantizer.fit(np.concatenate([h[0] for h in self._held]))
        self._index.quantizer = quantizer
        LOGGER.info("adding buffered vectors to index")
        held, self._held = self._held, []
        for batch in held:
            self._index.add(*batch)

    def _encode(self, texts: Sequence[str]) -> np.ndarray:
        if self._encoder is None:
            raise RuntimeError("An encoder is required.")
        step = self._encoder_batch_size
        return np.concatenate([self._encoder(texts[i:i + step]) for i in range(0, len(texts), step)])

    def from_dicts(self, data: Iterable[IndexingDict]) -> None:
        """Encode and index ``{"text": ..., "doc_id": ..., "psg_id": ...}`` records, ``batch_size`` at a time."""
        texts: list[str] = []
        docs: list = []
        psgs: list = []

        def flush() -> None:
            if texts:
                self._index_batch(self._encode(texts), doc_ids=list(docs), psg_ids=list(psgs))
                texts.clear()
                docs.clear()
                psgs.clear()

        for record in data:
            texts.append(record["text"])
            docs.append(record.get("doc_id"))
            psgs.append(record.get("psg_id"))
            if len(texts) == self._batch_size:
                flush()
        flush()

    def from_index(self, index) -> None:
        """Copy vectors (reconstructed, if the source is quantized) and ids from another index."""
        for vectors, doc_ids, psg_ids in index.batch_iter(self._batch_size):
            self._index_batch(vectors, doc_ids, psg_ids)
