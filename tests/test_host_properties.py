"""Property tests (hypothesis) of the host natives of libffx against their NumPy statements, at the small sizes where
edge cases live: empty inputs, single elements, all-equal values, ties, ragged query blocks."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

pytest.importorskip("fast_forward_amd.native")
from fast_forward_amd import native  # noqa: E402

small = settings(max_examples=120, deadline=None)


@small
@given(st.lists(st.integers(0, 40), min_size=0, max_size=12), st.integers(0, 2**31 - 1))
def test_host_order_property(lens, seed):
    """Every query's positions, scores descending, exact ties by position (-0.0 == +0.0)."""
    rng = np.random.default_rng(seed)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    s = rng.choice(np.array([-np.inf, -1.5, -0.0, 0.0, 0.25, 0.25, 3.0, np.inf], dtype=np.float32), size=int(off[-1]))
    got = native.host_order(s, off, 2)
    for q in range(len(lens)):
        block = got[off[q]:off[q + 1]]
        assert sorted(block.tolist()) == list(range(int(off[q]), int(off[q + 1])))
        keys = [(-float(s[p]) if s[p] != 0 else 0.0, int(p)) for p in block]
        assert keys == sorted(keys)


@small
@given(st.integers(1, 300), st.integers(0, 400), st.integers(0, 2**31 - 1), st.integers(1, 4))
def test_unique_rows_property(n_index, n, seed, threads):
    rows = np.random.default_rng(seed).integers(0, n_index, size=n).astype(np.int64)
    uniq, inverse = native.unique_rows(rows, n_index, threads)
    want_u, want_i = np.unique(rows, return_inverse=True)
    assert np.array_equal(uniq, want_u) and np.array_equal(inverse, want_i)


@small
@given(st.integers(1, 60), st.integers(0, 300), st.integers(0, 2**31 - 1), st.booleans())
def test_rows_unique_and_segments_property(n_docs, n, seed, two_tables):
    rng = np.random.default_rng(seed)
    per = rng.integers(1, 5, size=n_docs)
    entry_off = np.concatenate([[0], np.cumsum(per)]).astype(np.int32)
    n_index = int(entry_off[-1]) + int(rng.integers(0, 5))
    entry_row = rng.permutation(n_index)[:int(entry_off[-1])].astype(np.int32)
    ids = rng.integers(0, n_docs, size=n).astype(np.int32)
    # one row per candidate (FirstP: the document's first entry; or a single table)
    first = entry_row[entry_off[:-1][ids]] if two_tables else entry_off[:-1][ids]
    uniq, inverse = native.rows_unique(ids, np.ascontiguousarray(entry_off[:-1]), entry_row if two_tables else None, n_index, threads=2)
    want_u, want_i = np.unique(first, return_inverse=True)
    assert np.array_equal(uniq, want_u) and np.array_equal(inverse, want_i)
    # all rows of every candidate's document
    seg_off, uniq, inverse = native.segment_rows_unique(ids, entry_off, entry_row, n_index, threads=2)
    rows = np.concatenate([entry_row[entry_off[d]:entry_off[d + 1]] for d in ids]) if n else np.zeros(0, dtype=np.int32)
    assert np.array_equal(seg_off, np.concatenate([[0], np.cumsum(per[ids])]).astype(np.int64))
    want_u, want_i = np.unique(rows, return_inverse=True)
    assert np.array_equal(uniq, want_u) and np.array_equal(inverse, want_i)


@small
@given(st.lists(st.sampled_from(["", "a", "b", "D1", "D10", "d1", "ä", "a b", "D1 "]), min_size=1, max_size=40), st.integers(1, 4))
def test_string_numbering_depends_on_the_set_only(items, threads):
    """code_strings: equal strings share a number, the names decode back, and the numbering of a permutation of the
    same strings is the same numbering."""
    blob = "\n".join(items).encode()
    codes, names_blob, n_names = native.code_strings(blob, len(items), threads)
    names = names_blob.decode().split("\n") if n_names else []
    assert n_names == len(set(items)) and [names[c] for c in codes] == items
    again, names_blob2, _ = native.code_strings("\n".join(reversed(items)).encode(), len(items), threads)
    assert names_blob2 == names_blob and again.tolist() == codes.tolist()[::-1]


@small
@given(st.lists(st.integers(0, 30), min_size=0, max_size=200), st.integers(1, 4))
def test_code_u64_property(values, threads):
    """Numbers are dense, ascend with the values, and first_row names a row holding each number."""
    v = np.array(values, dtype=np.uint64) * np.uint64(48) + np.uint64(0x7F0000000000)
    codes, first = native.code_u64(v.ctypes.data, v.shape[0], threads) if v.shape[0] else (np.zeros(0, np.int32), np.zeros(0, np.int64))
    uniq = np.unique(v)
    assert first.shape[0] == uniq.shape[0]
    if v.shape[0]:
        assert np.array_equal(uniq[codes], v) and np.array_equal(v[first], uniq)


@small
@given(st.lists(st.tuples(st.integers(0, 40), st.integers(1, 9000)), min_size=1, max_size=60), st.integers(1, 4), st.booleans())
def test_code_u64_on_columns_of_long_runs_and_mixed_columns(runs, threads, sprinkle):
    """The run-head shortcut of ffx_host_code_u64 (a run's query-id column: thousands of rows in a row name one object) gives what
    the general scheme gives -- dense numbers ascending with the values, first_row naming a row of each -- for columns of long
    runs, columns where only a part is runs (a thread's range without long runs sends the whole column down the general path)
    and values that come back in later runs."""
    parts = [np.full(length, value, dtype=np.uint64) for value, length in runs]
    if sprinkle:  # a stretch without runs in the middle
        parts.insert(len(parts) // 2, np.arange(5000, dtype=np.uint64) % np.uint64(37) + np.uint64(100))
    v = np.concatenate(parts) * np.uint64(48) + np.uint64(0x7F0000000000)
    v = np.concatenate([v] * (1 + (1 << 16) // max(1, v.shape[0])))  # (the shortcut looks at columns of 65 536 rows and more)
    codes, first = native.code_u64(v.ctypes.data, v.shape[0], threads)
    uniq = np.unique(v)
    assert first.shape[0] == uniq.shape[0] and np.array_equal(uniq[codes], v) and np.array_equal(v[first], uniq)
