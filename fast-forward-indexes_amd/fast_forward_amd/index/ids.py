"""Host-side id tables: string document/passage ids -> integer ordinals -> index rows.

Replaces the two dicts of the reference's ``InMemoryIndex`` (``src/fast_forward/index/memory.py:46-47``)
and ``get_indices`` (``src/fast_forward/index/util.py:12-42``).  The device only ever sees
ordinals: a *document ordinal* (order of first appearance of the doc id) owns a list of rows
in insertion order, a *passage ordinal* (insertion order) owns one row.
"""

from __future__ import annotations

from collections.abc import Iterable, Sequence

import numpy as np
import pandas as pd

PASSAGE, MAXP, FIRSTP, AVEP = 1, 2, 3, 4


class IdTable:
    def __init__(self) -> None:
        self._doc_ordinal: dict[str, int] = {}
        self._doc_rows: list[list[int]] = []
        self._psg_ordinal: dict[str, int] = {}
        self._psg_row: list[int] = []
        self._csr: tuple[np.ndarray, np.ndarray] | None = None
        self._doc_lookup: pd.Index | None = None
        self._psg_lookup: pd.Index | None = None
        self._native: dict[bool, tuple[int, object]] = {}  # passage? -> (version, native IdStrings)
        self.version = 0  # bumped on every change; the device copy is refreshed lazily

    # ---- growth ------------------------------------------------------------------------
    def add(self, doc_ids: Sequence[str | None], psg_ids: Sequence[str | None], first_row: int) -> None:
        """Register rows ``first_row ..``.  A repeated passage id is a ``RuntimeError``
        (memory.py:93-94); nothing is modified in that case."""
        seen: set[str] = set()
        for p in psg_ids:
            if p is None:
                continue
            if p in self._psg_ordinal or p in seen:
                raise RuntimeError(f"Passage ID {p} already exists.")
            seen.add(p)
        for row, d in enumerate(doc_ids, first_row):
            if d is None:
                continue
            o = self._doc_ordinal.get(d)
            if o is None:
                o = len(self._doc_rows)
                self._doc_ordinal[d] = o
                self._doc_rows.append([])
            self._doc_rows[o].append(row)
        for row, p in enumerate(psg_ids, first_row):
            if p is None:
                continue
            self._psg_ordinal[p] = len(self._psg_row)
            self._psg_row.append(row)
        self._csr = self._doc_lookup = self._psg_lookup = None
        self.version += 1

    # ---- views -----------------------------------------------------------------------------
    @property
    def doc_ids(self) -> set[str]:
        return set(self._doc_ordinal)

    @property
    def psg_ids(self) -> set[str]:
        return set(self._psg_ordinal)

    def doc_csr(self) -> tuple[np.ndarray, np.ndarray]:
        """``(doc_off int64[D+1], doc_rows int32[T])``: rows of ordinal o are
        ``doc_rows[doc_off[o]:doc_off[o+1]]`` in insertion order."""
        if self._csr is None:
            counts = np.fromiter((len(r) for r in self._doc_rows), dtype=np.int64, count=len(self._doc_rows))
            off = np.zeros(len(self._doc_rows) + 1, dtype=np.int64)
            np.cumsum(counts, out=off[1:])
            rows = np.fromiter((r for rs in self._doc_rows for r in rs), dtype=np.int32, count=int(off[-1]))
            self._csr = (off, rows)
        return self._csr

    def psg_rows(self) -> np.ndarray:
        return np.asarray(self._psg_row, dtype=np.int32)

    def row_ids(self, n_rows: int) -> tuple[list[str | None], list[str | None]]:
        """Per-row (doc id, psg id), ``None`` where a row has none (for ``batch_iter``)."""
        docs: list[str | None] = [None] * n_rows
        psgs: list[str | None] = [None] * n_rows
        for d, o in self._doc_ordinal.items():
            for r in self._doc_rows[o]:
                docs[r] = d
        for p, o in self._psg_ordinal.items():
            psgs[self._psg_row[o]] = p
        return docs, psgs

    # ---- resolution ---------------------------------------------------------------------------
    def resolve(self, ids: np.ndarray | Sequence[str], mode: int) -> np.ndarray:
        """Ordinals (int32) of ``ids`` for ``mode``; ``IndexError`` names the first unknown id
        exactly like the reference (util.py:38-39)."""
        passage = mode == PASSAGE
        if passage:
            if self._psg_lookup is None:
                self._psg_lookup = pd.Index(list(self._psg_ordinal), dtype=object)
            lookup = self._psg_lookup
        else:
            if self._doc_lookup is None:
                self._doc_lookup = pd.Index(list(self._doc_ordinal), dtype=object)
            lookup = self._doc_lookup
        ids = np.asarray(ids, dtype=object)
        if len(ids) >= 4096:
            # large requests go through the native table (byte hashing in C, ~3x pandas)
            return self.native_table(mode).lookup(ids)
        codes, uniques = pd.factorize(ids)
        ordinals = lookup.get_indexer(uniques) if len(lookup) else np.full(len(uniques), -1, dtype=np.int64)
        if (ordinals < 0).any():
            first_bad = ids[np.flatnonzero(ordinals[codes] < 0)[0]]
            raise IndexError(f"ID {first_bad} not found in the index.")
        return ordinals[codes].astype(np.int32)

    def lookup_optional(self, ids, mode: int) -> np.ndarray:
        """Ordinals (int64) of ``ids`` for ``mode``, -1 where the id is not in the table."""
        table = self._psg_ordinal if mode == PASSAGE else self._doc_ordinal
        return np.fromiter((table.get(u, -1) for u in ids), dtype=np.int64, count=len(ids))

    def native_table(self, mode: int):
        """The native (libffx) mirror of the doc- or passage-id table, rebuilt when ids changed;
        ordinals agree with ``resolve``.  Used by the file-to-file path (``RunFile``)."""
        from fast_forward_amd import native

        passage = mode == PASSAGE
        cached = self._native.get(passage)
        if cached is None or cached[0] != self.version:
            table = native.IdStrings()
            keys = list(self._psg_ordinal if passage else self._doc_ordinal)
            got = table.intern(keys)
            assert len(keys) == 0 or (got[-1] == len(keys) - 1 and len(table) == len(keys)), "ids are not unique as bytes"
            self._native[passage] = cached = (self.version, table)
        return cached[1]

    def rows_for(self, ids: Iterable[str], mode: int) -> tuple[list[int], list[str]]:
        """``get_indices`` of the reference: the rows needed to score each id, and the id
        repeated once per row."""
        rows: list[int] = []
        owners: list[str] = []
        for i in ids:
            if mode == PASSAGE:
                o = self._psg_ordinal.get(i)
                cur = [] if o is None else [self._psg_row[o]]
            else:
                o = self._doc_ordinal.get(i)
                cur = [] if o is None else self._doc_rows[o]
                if mode == FIRSTP:
                    cur = cur[:1]
            if not cur:
                raise IndexError(f"ID {i} not found in the index.")
            rows.extend(cur)
            owners.extend([i] * len(cur))
        return rows, owners
