"""Host-side id tables: string document/passage ids -> integer ordinals -> index rows.

Replaces the two dicts of the reference's ``InMemoryIndex`` (``src/fast_forward/index/memory.py:46-47``)
and ``get_indices`` (``src/fast_forward/index/util.py:12-42``).  The device only ever sees
ordinals: a *document ordinal* (order of first appearance of the doc id) owns a list of rows
in insertion order, a *passage ordinal* (insertion order) owns one row.

The strings live in libffx's interning tables (``ffx_ids``: open addressing over one byte arena) and
everything per row is a NumPy array: ``add`` of millions of ids is two hashing passes in C instead
of a Python loop with a dict insert and a list append per row (the reference: ~24 us per vector,
memory.py:85-99), and a look-up never touches a Python string object.
"""

from __future__ import annotations

from collections.abc import Iterable, Sequence

import numpy as np

PASSAGE, MAXP, FIRSTP, AVEP = 1, 2, 3, 4


def _present(ids: Sequence["str | None"]) -> tuple[np.ndarray | None, list | np.ndarray]:
    """(positions that hold an id, or None when every position does; the ids at those positions)."""
    if isinstance(ids, np.ndarray) and ids.dtype.kind in "US":
        return None, ids
    arr = np.empty(len(ids), dtype=object)
    arr[:] = ids
    missing = np.equal(arr, None)
    if not missing.any():
        return None, arr
    where = np.flatnonzero(~missing)
    return where, arr[where]


class IdTable:
    def __init__(self) -> None:
        from fast_forward_amd import native

        self._docs = native.IdStrings()   # document id -> ordinal (order of first appearance)
        self._psgs = native.IdStrings()   # passage id  -> ordinal (insertion order)
        self._row_doc: list[np.ndarray] = []   # per add(): document ordinal of every row (-1: none)
        self._psg_row: list[np.ndarray] = []   # per add(): row of every new passage ordinal
        self._n_rows = 0
        self._csr: tuple[np.ndarray, np.ndarray] | None = None
        self._psg_rows_all: np.ndarray | None = None
        self.version = 0  # bumped on every change; the device copy is refreshed lazily

    # ---- growth ------------------------------------------------------------------------
    def add(self, doc_ids: Sequence[str | None], psg_ids: Sequence[str | None], first_row: int) -> None:
        """Register rows ``first_row ..``.  A repeated passage id is a ``RuntimeError``
        (memory.py:93-94); nothing is modified in that case."""
        from fast_forward_amd import native

        n = len(doc_ids)
        p_where, p_strs = _present(psg_ids)
        if len(p_strs):
            known = self._psgs.lookup(p_strs, missing_ok=True) if len(self._psgs) else np.full(len(p_strs), -1, dtype=np.int32)
            scratch = native.IdStrings()
            repeated = scratch.intern(p_strs) != np.arange(len(p_strs), dtype=np.int32)  # a second occurrence inside this call
            scratch.close()
            bad = np.flatnonzero((known >= 0) | repeated)
            if bad.size:
                raise RuntimeError(f"Passage ID {p_strs[int(bad[0])]} already exists.")
        d_where, d_strs = _present(doc_ids)
        row_doc = np.full(n, -1, dtype=np.int32)
        if len(d_strs):
            ordinals = self._docs.intern(d_strs)
            if d_where is None:
                row_doc = ordinals
            else:
                row_doc[d_where] = ordinals
        if len(p_strs):
            self._psgs.intern(p_strs)
            rows = np.arange(first_row, first_row + n, dtype=np.int32)
            self._psg_row.append(rows if p_where is None else rows[p_where])
        self._row_doc.append(row_doc)
        assert first_row == self._n_rows, "rows are registered in insertion order"
        self._n_rows += n
        self._csr = self._psg_rows_all = None
        self.version += 1

    # ---- views -----------------------------------------------------------------------------
    @property
    def n_docs(self) -> int:
        return len(self._docs)

    @property
    def n_psgs(self) -> int:
        return len(self._psgs)

    @property
    def doc_ids(self) -> set[str]:
        return set(self._docs.strings())

    @property
    def psg_ids(self) -> set[str]:
        return set(self._psgs.strings())

    def _row_doc_all(self) -> np.ndarray:
        if len(self._row_doc) > 1:
            self._row_doc = [np.concatenate(self._row_doc)]
        return self._row_doc[0] if self._row_doc else np.zeros(0, dtype=np.int32)

    def doc_csr(self) -> tuple[np.ndarray, np.ndarray]:
        """``(doc_off int64[D+1], doc_rows int32[T])``: rows of ordinal o are
        ``doc_rows[doc_off[o]:doc_off[o+1]]`` in insertion order."""
        if self._csr is None:
            row_doc = self._row_doc_all()
            rows = np.flatnonzero(row_doc >= 0).astype(np.int32) if (row_doc < 0).any() else np.arange(row_doc.shape[0], dtype=np.int32)
            owner = row_doc[rows]
            off = np.zeros(self.n_docs + 1, dtype=np.int64)
            np.cumsum(np.bincount(owner, minlength=self.n_docs), out=off[1:])
            if owner.shape[0] > 1 and (np.diff(owner) < 0).any():  # a document re-opened later: group its rows (stable)
                rows = rows[np.argsort(owner, kind="stable")]
            self._csr = (off, rows)
        return self._csr

    def psg_rows(self) -> np.ndarray:
        if self._psg_rows_all is None:
            self._psg_rows_all = np.concatenate(self._psg_row).astype(np.int32) if self._psg_row else np.zeros(0, dtype=np.int32)
        return self._psg_rows_all

    def row_ids(self, n_rows: int) -> tuple[list[str | None], list[str | None]]:
        """Per-row (doc id, psg id), ``None`` where a row has none (for ``batch_iter``)."""
        names = np.empty(self.n_docs + 1, dtype=object)  # the extra slot is None: rows without a document
        names[:self.n_docs] = self._docs.strings()
        row_doc = self._row_doc_all()[:n_rows]
        docs = names[np.where(row_doc >= 0, row_doc, self.n_docs)].tolist() + [None] * max(0, n_rows - row_doc.shape[0])
        psgs = np.empty(n_rows, dtype=object)
        rows = self.psg_rows()
        inside = rows < n_rows
        p_names = np.empty(self.n_psgs, dtype=object)
        p_names[:] = self._psgs.strings()
        psgs[rows[inside]] = p_names[inside]
        return docs, psgs.tolist()

    # ---- resolution ---------------------------------------------------------------------------
    def resolve(self, ids: np.ndarray | Sequence[str], mode: int) -> np.ndarray:
        """Ordinals (int32) of ``ids`` for ``mode``; ``IndexError`` names the first unknown id
        exactly like the reference (util.py:38-39)."""
        return self.native_table(mode).lookup(ids if isinstance(ids, np.ndarray) else np.asarray(ids, dtype=object))

    def lookup_optional(self, ids, mode: int, packed=None) -> np.ndarray:
        """Ordinals (int64) of ``ids`` for ``mode``, -1 where the id is not in the table.  ``packed``: the ids once more
        as (one ``'\n'``-joined UTF-8 buffer, their number), when the caller holds them that way."""
        if packed is None or packed[1] != len(ids):  # (with the joined buffer at hand no string object is looked at -- or made: a
            # ranking's ``Names`` would turn into millions of Python strings here, 0.3 s for the 8.7 M ids of the headline run)
            ids = ids if isinstance(ids, np.ndarray) else np.asarray(ids, dtype=object)
        return self.native_table(mode).lookup(ids, missing_ok=True, packed=packed).astype(np.int64)

    def native_table(self, mode: int):
        """The native (libffx) doc- or passage-id table; ordinals agree with ``resolve``.  Also used by
        the file-to-file path (``RunFile``)."""
        return self._psgs if mode == PASSAGE else self._docs

    def rows_for(self, ids: Iterable[str], mode: int) -> tuple[list[int], list[str]]:
        """``get_indices`` of the reference: the rows needed to score each id, and the id
        repeated once per row."""
        ids = list(ids)
        if not ids:
            return [], []
        ordinals = self.resolve(np.asarray(ids, dtype=object), mode).astype(np.int64)
        if mode == PASSAGE:
            return self.psg_rows()[ordinals].tolist(), ids
        off, doc_rows = self.doc_csr()
        rows: list[int] = []
        owners: list[str] = []
        for i, o in zip(ids, ordinals.tolist()):
            cur = doc_rows[off[o]:off[o + 1]].tolist()
            if mode == FIRSTP:
                cur = cur[:1]
            rows.extend(cur)
            owners.extend([i] * len(cur))
        return rows, owners
