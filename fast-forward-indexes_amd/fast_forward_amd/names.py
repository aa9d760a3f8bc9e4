"""``Names``: the table of distinct id strings of a coded ``Ranking``, held as ONE buffer.

A run of 34.9 M rows names 8.7 M documents.  The device path needs those names twice -- to look them up in the index's id
table and to print them into a run file -- and both natives take them as one ``'\\n'``-joined UTF-8 buffer
(``ffx_ids_lookup_packed``, ``ffx_run_save_coded``).  Making 8.7 M Python strings out of that buffer (2 s) just to have them
is what ``Ranking.from_file`` used to spend most of its time on; here they are made if and when somebody asks for one
(an error message, the pandas frame of a result) -- the reference's frame holds a Python string per ROW
(``src/fast_forward/ranking.py:95-116``), this holds none until then.

Behaves like the read-only object array it replaces where the package indexes it: ``len``, ``names[i]``,
``names[int_array]``, ``np.asarray(names)``, iteration."""

from __future__ import annotations

import numpy as np


class Names:
    __slots__ = ("blob", "n", "_objects", "_starts", "_plain", "__weakref__")

    def __init__(self, blob: bytes, n: int, objects: "np.ndarray | None" = None) -> None:
        """``n`` strings joined by ``'\\n'`` in ``blob`` (no string holds one); ``objects``: the same strings as an
        object array, if the caller has them anyway."""
        self.blob, self.n = blob, int(n)
        self._objects = objects
        self._starts = None
        self._plain = None

    # ---- what the natives take ----------------------------------------------------------------------------
    @property
    def packed(self) -> "tuple[bytes, int] | None":
        return (self.blob, self.n) if self.n else None

    def plain_for_csv(self) -> bool:
        """Whether a CSV writer would print every name as it is (no separator, quote or line-break character inside, no
        empty name): then a run file can be written from the buffer directly."""
        if self._plain is None:
            b = self.blob
            self._plain = not (any(c in b for c in (b"\t", b'"', b"\r", b"\0")) or (self.n and (b.startswith(b"\n") or b.endswith(b"\n") or b"\n\n" in b or not b)))
        return self._plain

    # ---- the array it stands for --------------------------------------------------------------------------
    def __len__(self) -> int:
        return self.n

    @property
    def shape(self) -> tuple[int]:
        return (self.n,)

    ndim = 1
    dtype = np.dtype(object)

    def objects(self) -> np.ndarray:
        """The strings as an object array (made once, then kept)."""
        if self._objects is None:
            arr = np.empty(self.n, dtype=object)
            if self.n:
                arr[:] = self.blob.decode("utf-8").split("\n")
            self._objects = arr
        return self._objects

    def __array__(self, dtype=None, copy=None):
        arr = self.objects()
        return arr if dtype is None or np.dtype(dtype) == object else arr.astype(dtype)

    def _one(self, i: int) -> str:
        if self._objects is not None:
            return self._objects[i]
        if self._starts is None:  # where every string starts: one vectorised scan, only when a single name is asked for
            seps = np.flatnonzero(np.frombuffer(self.blob, dtype=np.uint8) == 10)
            self._starts = np.concatenate([[0], seps + 1, [len(self.blob) + 1]]).astype(np.int64)
        n = self.n
        if not -n <= i < n:
            raise IndexError(f"index {i} is out of bounds for {n} names")
        i %= n
        return self.blob[int(self._starts[i]):int(self._starts[i + 1]) - 1].decode("utf-8")

    def __getitem__(self, key):
        if isinstance(key, (int, np.integer)):
            return self._one(int(key))
        return self.objects()[key]

    def __iter__(self):
        return iter(self.objects())

    def tolist(self) -> list:
        return self.objects().tolist()

    @classmethod
    def joined(cls, a: "Names", b: "Names") -> "Names":
        """The strings of ``a`` followed by those of ``b`` (not made distinct)."""
        if a.n == 0 or b.n == 0:
            return b if a.n == 0 else a
        return cls(a.blob + b"\n" + b.blob, a.n + b.n)
