"""The native host side of libffx (no GPU needed): string-id tables and TREC run files against
what ``Ranking.from_file`` / ``cut`` / ``save`` do (reference ranking.py:279-291, :348-409)."""

import numpy as np
import pandas as pd
import pytest

from fast_forward_amd import Ranking, native


def test_id_table_interning_and_lookup():
    t = native.IdStrings()
    assert t.intern(["d0", "d1", "d0", "doc with spaces", "d2", "d1"]).tolist() == [0, 1, 0, 2, 3, 1]
    assert len(t) == 4 and t.get(2) == "doc with spaces"
    assert t.lookup(np.array(["d2", "d0"])).tolist() == [3, 0]
    assert t.lookup(np.array([b"d1"], dtype="S8")).tolist() == [1]
    with pytest.raises(IndexError, match="ID nope not found in the index."):
        t.lookup(["d0", "nope", "also-no"])
    # growth across many rehashes keeps ordinals stable
    many = [f"id-{i}" for i in range(50_000)]
    got = t.intern(many)
    assert got[0] == 4 and got[-1] == 4 + len(many) - 1 and t.lookup(["id-31415"]).tolist() == [4 + 31415]
    assert t.intern([]).shape == (0,) and t.lookup([]).shape == (0,)


RUN_TEXT = """q1 Q0 d0 1 1.5 bm25
q1 Q0 d2 2 300 bm25
q2\tQ0\td3\t1\t7.25\tbm25
q1   Q0  d1  3  2e0  bm25

q2 Q0 d0 2 4 bm25
q2 Q0 d2 3 600.125 bm25
q2 Q0 d1 4 5 bm25
"""


@pytest.fixture
def run_path(tmp_path):
    p = tmp_path / "run.tsv"
    p.write_text(RUN_TEXT)
    return p


def _table(*ids):
    t = native.IdStrings()
    t.intern(list(ids))
    return t


def test_run_parse_matches_ranking_from_file(run_path):
    table = _table("d0", "d1", "d2", "d3")
    run = native.RunFile(run_path, table)
    assert (run.n_queries, run.n_pairs, run.tag) == (2, 7, "bm25")
    assert run.query_ids() == ["q1", "q2"]
    pair_off, pair_ids, sparse = run.export()
    assert pair_off.tolist() == [0, 3, 7]
    assert pair_ids.tolist() == [0, 2, 1, 3, 0, 2, 1]  # grouped by query, file order inside
    assert sparse.tolist() == [1.5, 300.0, 2.0, 7.25, 4.0, 600.125, 5.0]
    ref = Ranking.from_file(run_path)
    for qi, q in enumerate(run.query_ids()):
        got = {table.get(o): s for o, s in zip(pair_ids[pair_off[qi]:pair_off[qi + 1]], sparse[pair_off[qi]:pair_off[qi + 1]])}
        assert got == ref[q]
    run.close()


def test_run_cut_keeps_the_best_of_each_query(run_path):
    table = _table("d0", "d1", "d2", "d3")
    run = native.RunFile(run_path, table)
    pair_off, pair_ids, sparse = run.export(cutoff=2)
    assert pair_off.tolist() == [0, 2, 4] and run.n_pairs == 4
    ref = Ranking.from_file(run_path).cut(2)
    for qi, q in enumerate(run.query_ids()):
        got = {table.get(o): s for o, s in zip(pair_ids[pair_off[qi]:pair_off[qi + 1]], sparse[pair_off[qi]:pair_off[qi + 1]])}
        assert got == ref[q]


def test_run_errors(run_path, tmp_path):
    with pytest.raises(IndexError, match="ID d3 not found in the index."):
        native.RunFile(run_path, _table("d0", "d1", "d2"))
    dup = tmp_path / "dup.tsv"
    dup.write_text("q1 Q0 d0 1 1.0 x\nq1 Q0 d0 2 0.5 x\n")
    with pytest.raises(ValueError, match="Only one score per query-document/passage pair"):
        native.RunFile(dup, _table("d0"))
    short = tmp_path / "short.tsv"
    short.write_text("q1 Q0 d0\n")
    with pytest.raises(ValueError):
        native.RunFile(short, _table("d0"))
    with pytest.raises(ValueError):
        native.RunFile(tmp_path / "missing.tsv", _table("d0"))


def test_run_write_is_byte_identical_to_ranking_save(run_path, tmp_path):
    """Same rows, ranks and float formatting as pandas' to_csv of a float32 column."""
    table = _table("d0", "d1", "d2", "d3")
    run = native.RunFile(run_path, table)
    pair_off, pair_ids, _ = run.export()
    scores = np.array([152.0, 3.5, 1e-5, 1234567.0, 100000.0, 0.1, 1e16], dtype=np.float32)
    scores[3] = np.float32(-3.25e-7)
    # order: per query, descending score, ties by position (what the device sort returns)
    order = np.concatenate([
        pair_off[q] + np.argsort(-scores[pair_off[q]:pair_off[q + 1]].astype(np.float64), kind="stable") for q in range(run.n_queries)
    ]).astype(np.int32)
    out_native = tmp_path / "native.tsv"
    run.write(out_native, scores, order, tag="fused")
    q_names = run.query_ids()
    df = pd.DataFrame({
        "q_id": np.repeat(q_names, np.diff(pair_off)), "id": [table.get(o) for o in pair_ids], "score": scores})
    out_py = tmp_path / "python.tsv"
    Ranking(df, name="fused").save(out_py)
    assert out_native.read_text() == out_py.read_text()
    # keep=2 writes the top two of every query
    run.write(tmp_path / "top2.tsv", scores, order, keep=2, tag="fused")
    assert (tmp_path / "top2.tsv").read_text() == "".join(
        line + "\n" for line in out_py.read_text().splitlines() if int(line.split("\t")[3]) <= 2)


@pytest.mark.parametrize("value", [0.0, -0.0, 1.0, 152.0, 0.1, 1e-5, 9.999e-5, 1e-4, 123456.7, 1e15, 1e16, 3.4e38, -2.5e-38, 16777216.0])
def test_float_formatting_matches_numpy_str(value, tmp_path):
    table = _table("d0")
    src = tmp_path / "one.tsv"
    src.write_text("q Q0 d0 1 0 t\n")
    run = native.RunFile(src, table)
    run.write(tmp_path / "o.tsv", np.array([value], dtype=np.float32), np.array([0], dtype=np.int32))
    assert (tmp_path / "o.tsv").read_text().split("\t")[4] == str(np.float32(value))


def test_lookup_of_python_strings_goes_through_one_packed_buffer():
    """Large object arrays are joined into one buffer and split natively (ffx_ids_lookup_packed,
    threaded): same ordinals as the fixed-width path, same first missing id, odd strings fall back."""
    table = native.IdStrings()
    known = [f"D{i}" for i in range(50_000)] + ["dóc-ü", "x" * 70, ""]
    table.intern(np.array(known, dtype=object))
    rng = np.random.default_rng(3)
    pick = rng.integers(0, len(known), size=200_000)
    want = np.asarray(known, dtype=object)[pick]
    got = table.lookup(want)
    assert np.array_equal(got, pick.astype(np.int32))
    assert np.array_equal(got, table.lookup(native._as_fixed_bytes(want)))  # the fixed-width path
    bad = want.copy()
    bad[150_001] = "nope"
    bad[199_000] = "neither"
    with pytest.raises(IndexError, match="ID nope not found"):
        table.lookup(bad)
    odd = want.copy()
    odd[17] = "two\nlines"  # cannot be packed with a newline separator: falls back, then fails as unknown
    with pytest.raises(IndexError, match="two"):
        table.lookup(odd)
    table.close()


def test_large_run_is_parsed_and_written_in_parallel_pieces(tmp_path):
    """A file of several MiB is cut into pieces parsed by threads (and written back in batches):
    numbering, CSR, errors with their line numbers and the written bytes are those of a serial pass."""
    rng = np.random.default_rng(4)
    n_docs, n_q, k = 20_000, 300, 1000
    table = _table(*[f"D{i}" for i in range(n_docs)])
    q_names = [f"q{i:04d}" for i in rng.permutation(n_q)]  # queries not in sorted order in the file
    lines, want_ids, want_scores = [], [], []
    for q in q_names:
        docs = rng.choice(n_docs, size=k, replace=False)
        scores = rng.random(k).astype(np.float32) * 30
        want_ids.append(docs)
        want_scores.append(scores)
        lines += [f"{q} Q0 D{d} {r + 1} {s} first" for r, (d, s) in enumerate(zip(docs, scores))]
    path = tmp_path / "big.tsv"
    path.write_text("\n".join(lines) + "\n")
    assert path.stat().st_size > 8 << 20
    run = native.RunFile(path, table)
    assert (run.n_queries, run.n_pairs, run.tag) == (n_q, n_q * k, "first")
    assert run.query_ids() == q_names
    pair_off, pair_ids, sparse = run.export()
    assert np.array_equal(pair_off, np.arange(n_q + 1) * k)
    assert np.array_equal(pair_ids, np.concatenate(want_ids)) and np.array_equal(sparse, np.concatenate(want_scores))
    # written bytes == Ranking.save of the same scores
    new = rng.random(n_q * k).astype(np.float32)
    order = np.concatenate([q * k + np.argsort(-new[q * k:(q + 1) * k].astype(np.float64), kind="stable") for q in range(n_q)]).astype(np.int32)
    run.write(tmp_path / "native.tsv", new, order, tag="second")
    df = pd.DataFrame({"q_id": np.repeat(q_names, k), "id": [f"D{d}" for d in pair_ids], "score": new})
    Ranking(df, name="second").save(tmp_path / "python.tsv")
    assert (tmp_path / "native.tsv").read_bytes() == (tmp_path / "python.tsv").read_bytes()
    run.close()
    # problems deep inside the file: the first one in file order wins, with its line number
    broken = list(lines)
    broken[250_000] = broken[250_000].replace(" Q0 D", " Q0 X")
    broken[120_000] = "q0001 Q0 D5 1 1.5abc first"
    path.write_text("\n".join(broken) + "\n")
    with pytest.raises(ValueError, match=r":120001: bad score '1.5abc'"):
        native.RunFile(path, table)
    broken[120_000] = lines[120_000]
    path.write_text("\n".join(broken) + "\n")
    with pytest.raises(IndexError, match=r"ID X\d+ not found in the index."):
        native.RunFile(path, table)
    # a repeated pair far apart in the file
    broken[250_000] = lines[250_000]
    q_of = broken[10].split()[0]
    broken.append(f"{q_of} Q0 {broken[10].split()[2]} 9 0.5 first")
    path.write_text("\n".join(broken) + "\n")
    with pytest.raises(ValueError, match="Only one score per query-document/passage pair"):
        native.RunFile(path, table)


def test_run_parser_drops_rows_with_missing_scores_and_refuses_non_decimal_ones(tmp_path):
    """File-to-file path: a score that is one of pandas' NA words drops the row (Ranking's dropna,
    ranking.py:103); a spelling pandas would not read as a number is an error, not a silent number."""
    table = _table("d0", "d1", "d2")
    p = tmp_path / "na.tsv"
    p.write_text("q1 Q0 d0 1 1.5 t\nq1 Q0 d1 2 nan t\nq1 Q0 d2 3 NA t\nq2 Q0 d1 1 N/A t\nq3 Q0 d2 1 inf t\n")
    run = native.RunFile(p, table)
    pair_off, pair_ids, sparse = run.export()
    assert run.query_ids() == ["q1", "q3"] and pair_off.tolist() == [0, 1, 2] and pair_ids.tolist() == [0, 2]
    assert sparse.tolist() == [1.5, float("inf")]
    for bad in ("0x1A", "0x1p3", "nan(123)", "NAN", "1e", "abc"):
        p.write_text(f"q1 Q0 d0 1 1.5 t\nq1 Q0 d1 2 {bad} t\n")
        with pytest.raises(ValueError, match="bad score"):
            native.RunFile(p, table)


def test_rows_unique_equals_numpy():
    """ffx_host_rows_unique: table look-ups + numpy.unique(return_inverse=True) in one native pass."""
    rng = np.random.default_rng(3)
    t1 = rng.integers(0, 5000, size=4000).astype(np.int32)
    t2 = rng.integers(0, 9000, size=5000).astype(np.int32)
    for n in (0, 1, 300, 200_000):
        ids = rng.integers(0, 4000, size=n).astype(np.int32)
        for tables, rows, n_index in (((t1, t2), t2[t1[ids]], 9000), ((t1, None), t1[ids], 5000)):
            uniq, inverse = native.rows_unique(ids, tables[0], tables[1], n_index, threads=3)
            want_u, want_i = np.unique(rows, return_inverse=True)
            assert np.array_equal(uniq, want_u) and np.array_equal(inverse, want_i)
    with pytest.raises(native.FFXError, match="names no row"):
        native.rows_unique(np.array([4000], dtype=np.int32), t1, t2, 9000)
    with pytest.raises(native.FFXError, match="names no row"):
        native.rows_unique(np.array([0], dtype=np.int32), t1, t2, int(t2[t1[0]]))


def test_segment_rows_unique_equals_numpy():
    """ffx_host_segment_offsets + ffx_host_segment_rows_unique: every row of every candidate's document, the distinct
    rows and the slots, against the NumPy statement of get_indices (index/util.py:29-41) + numpy.unique."""
    rng = np.random.default_rng(4)
    n_docs = 3000
    per = rng.integers(1, 7, size=n_docs)
    entry_off = np.concatenate([[0], np.cumsum(per)]).astype(np.int32)
    entry_row = rng.permutation(int(entry_off[-1]) + 50)[:int(entry_off[-1])].astype(np.int32)
    n_index = int(entry_off[-1]) + 50
    for n in (0, 1, 700, 150_000):
        ids = rng.integers(0, n_docs, size=n).astype(np.int32)
        seg_off, uniq, inverse = native.segment_rows_unique(ids, entry_off, entry_row, n_index, threads=3)
        start = entry_off[ids].astype(np.int64)
        cnt = entry_off[ids.astype(np.int64) + 1] - start
        want_off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
        rows = entry_row[np.repeat(start, cnt) + np.arange(int(want_off[-1])) - np.repeat(want_off[:-1], cnt)] if n else np.zeros(0, dtype=np.int32)
        want_u, want_i = np.unique(rows, return_inverse=True)
        assert np.array_equal(seg_off, want_off) and np.array_equal(uniq, want_u) and np.array_equal(inverse, want_i)
    with pytest.raises(native.FFXError, match="names no document"):
        native.segment_rows_unique(np.array([n_docs], dtype=np.int32), entry_off, entry_row, n_index)
    with pytest.raises(native.FFXError, match="outside the index"):
        native.segment_rows_unique(np.array([0], dtype=np.int32), entry_off, entry_row, int(entry_row[0]))


def test_host_natives_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """csrc/host_io.cpp -- the code that parses untrusted run files -- built alone with
    g++ -fsanitize=address,undefined and driven by tests/c_abi/host_io_fuzz.cpp: a well-formed run,
    20 hand-written adversarial files (truncated, NUL bytes, 64 KiB fields, hex floats, missing
    columns, not UTF-8 ...) and 3000 seeded mutations of them through every ffx_ids_* / ffx_run_*
    entry point.  Any out-of-bounds access or undefined behaviour aborts the driver."""
    import shutil
    import subprocess
    from pathlib import Path

    if shutil.which("g++") is None:
        pytest.skip("no g++ here")
    csrc = Path(__file__).resolve().parent.parent / "fast-forward-indexes_amd" / "csrc"
    build = subprocess.run(["make", "-C", str(csrc), "host_asan"], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stdout[-2000:] + build.stderr[-4000:]
    run = subprocess.run([str(csrc / "build" / "host_io_asan"), str(tmp_path), "3000"], capture_output=True, text=True, timeout=600,
                         env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1", "PATH": "/usr/bin:/bin"})
    assert run.returncode == 0, run.stdout[-1000:] + run.stderr[-6000:]
    assert "host_io sanitizer run ok" in run.stdout


def test_host_natives_under_thread_sanitizer(tmp_path):
    """The same driver built with g++ -fsanitize=thread: every threaded native of csrc/host_io.cpp (order, gathers,
    interpolate, distinct rows, pointer and string numbering, crc32 owners, outer join) runs with four threads at sizes where
    the pools really start several, and must give the single-thread answer.  A data race aborts the driver."""
    import shutil
    import subprocess
    from pathlib import Path

    if shutil.which("g++") is None:
        pytest.skip("no g++ here")
    csrc = Path(__file__).resolve().parent.parent / "fast-forward-indexes_amd" / "csrc"
    build = subprocess.run(["make", "-C", str(csrc), "host_tsan"], capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stdout[-2000:] + build.stderr[-4000:]
    run = subprocess.run([str(csrc / "build" / "host_io_tsan"), str(tmp_path), "200"], capture_output=True, text=True, timeout=900,
                         env={"TSAN_OPTIONS": "halt_on_error=1:exitcode=66", "PATH": "/usr/bin:/bin"})
    assert run.returncode == 0, run.stdout[-1000:] + run.stderr[-6000:]
    assert "host_io sanitizer run ok" in run.stdout


def test_host_order_equals_the_oracle_order():
    """ffx_host_order (threads over queries, 64-bit words of inverted score image and position): scores
    descending, exact ties by position, -0.0 == +0.0, infinities -- the order the device's FFX_FLAG_SORT gives."""
    import ffx_oracle as fo

    rng = np.random.default_rng(0)
    lens = rng.integers(0, 3000, size=300)
    lens[5] = 0
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    s = rng.integers(-5, 6, size=int(off[-1])).astype(np.float32)
    s[::7] *= np.float32(0.37)
    zeros = np.flatnonzero(s == 0)
    s[zeros[::2]] = -0.0
    s[rng.integers(0, len(s), 5)] = np.inf
    s[rng.integers(0, len(s), 5)] = -np.inf
    for threads in (0, 1, 3):
        assert np.array_equal(native.host_order(s, off, threads).astype(np.int64), fo.order(s, off))
    assert native.host_order(np.zeros(0, dtype=np.float32), np.array([0, 0, 0])).shape == (0,)
    with pytest.raises(native.FFXError):
        native.host_order(s, np.array([0, 5, 3]))


def test_native_take_equals_numpy_take():
    """ffx_host_take32 (threads over blocks of the output) against NumPy's fancy indexing, for both element
    types it is used with; an index outside the source is refused; small inputs take NumPy's route."""
    from fast_forward_amd import native

    rng = np.random.default_rng(5)
    n_src, n = 700_001, 1_300_007
    index = rng.integers(0, n_src, size=n).astype(np.int32)
    for src in (rng.standard_normal(n_src).astype(np.float32), rng.integers(-2**31, 2**31 - 1, size=n_src).astype(np.int32)):
        for threads in (0, 1, 3):
            got = native.take32(src, index, threads)
            assert got.dtype == src.dtype and np.array_equal(got.view(np.uint32), src[index].view(np.uint32))
    bad = index.copy()
    bad[n // 2] = n_src
    with pytest.raises(native.FFXError) as e:
        native.take32(src, bad)
    assert f"index[{n // 2}]" in str(e.value)
    bad[n // 2] = -1
    with pytest.raises(native.FFXError):
        native.take32(src, bad)
    small = index[:1000]
    assert np.array_equal(native.take32(src, small), src[small])
    assert np.array_equal(native.take32(src.astype(np.float64), index), src.astype(np.float64)[index])  # not 4-byte: NumPy


def test_native_interpolate_equals_numpy_expression():
    """ffx_host_interpolate against the NumPy statement of ranking.py:319 on float32 arrays (scatter the dense
    scores back, then alpha * s + (1 - alpha) * d with float32-rounded weights, no fused multiply-add):
    bit-identical; NaN is reported; a source_row that is not a permutation is refused."""
    from fast_forward_amd import native

    rng = np.random.default_rng(9)
    n = 900_001
    sparse = (rng.standard_normal(n) * 17).astype(np.float32)
    dense_of = (rng.standard_normal(n) * 3).astype(np.float32)
    link = rng.permutation(n).astype(np.int32)
    for alpha in (0.1, 0.5, 0.0, 1.0, 0.3333333333333333):
        dense = np.empty_like(sparse)
        dense[link] = dense_of
        want = alpha * sparse + (1 - alpha) * dense
        assert want.dtype == np.float32
        for threads in (0, 1, 5):
            got, any_nan = native.host_interpolate(sparse, dense_of, link, alpha, threads)
            assert not any_nan and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    with_nan = dense_of.copy()
    with_nan[12345] = np.nan
    assert native.host_interpolate(sparse, with_nan, link, 0.2)[1]
    twice = link.copy()
    twice[5] = twice[6]
    with pytest.raises(native.FFXError, match="not a permutation"):
        native.host_interpolate(sparse, dense_of, twice, 0.2)
    outside = link.copy()
    outside[7] = n
    with pytest.raises(native.FFXError, match="not a permutation"):
        native.host_interpolate(sparse, dense_of, outside, 0.2)


def test_strings_are_numbered_like_a_factorisation():
    """ffx_strings_code: equal strings get equal codes, codes are dense, the returned names decode every row
    (the contract of pandas.factorize that Ranking relies on), for any thread count -- and the numbering itself
    does not depend on the thread count.  Empty strings, non-ASCII, one string, everything equal."""
    from fast_forward_amd import native

    rng = np.random.default_rng(3)
    for n, pool in ((1, 1), (5, 2), (70_000, 1), (100_000, 3000), (300_000, 250_000)):
        names = [f"D{v}" if v % 7 else (f"ü{v}" if v % 2 else "") for v in rng.integers(0, pool, size=n)]
        blob = "\n".join(names).encode()
        ref = None
        for threads in (1, 3, 8):
            codes, joined, k = native.code_strings(blob, n, threads)
            distinct = joined.decode().split("\n") if k else []
            assert len(distinct) == k == len(set(names)) == len(set(distinct))
            assert codes.min() >= 0 and codes.max() == k - 1
            assert [distinct[c] for c in codes] == names
            if ref is None:
                ref = codes.copy()
            assert np.array_equal(ref, codes)
    assert native.code_strings(b"", 0)[2] == 0
    with pytest.raises(native.FFXError, match="separated strings"):
        native.code_strings(b"a\nb", 3)


def test_coded_run_writer_equals_the_string_writer(tmp_path):
    """ffx_run_save_coded (rows hold codes into a table of names) writes the bytes ffx_run_save writes for the
    same rows spelled out; a code outside the table is refused."""
    from fast_forward_amd import native

    rng = np.random.default_rng(4)
    n_q, k, pool = 37, 300, 1000
    names = np.array([f"doc-{i}" for i in range(pool)], dtype=object)
    codes = np.concatenate([rng.choice(pool, size=k, replace=False) for _ in range(n_q)]).astype(np.int32)
    scores = (rng.standard_normal(n_q * k) * 7).astype(np.float32)
    pair_off = np.arange(n_q + 1, dtype=np.int64) * k
    q_names = [f"q{i}" for i in range(n_q)]
    assert native.save_run_coded(tmp_path / "coded.tsv", q_names, pair_off, codes, names, scores, "tag")
    assert native.save_run(tmp_path / "plain.tsv", q_names, pair_off, names[codes], scores, "tag")
    assert (tmp_path / "coded.tsv").read_bytes() == (tmp_path / "plain.tsv").read_bytes()
    assert native.save_run_coded(tmp_path / "blob.tsv", q_names, pair_off, codes, ("\n".join(names).encode(), pool), scores, "tag")
    assert (tmp_path / "blob.tsv").read_bytes() == (tmp_path / "plain.tsv").read_bytes()
    bad = codes.copy()
    bad[17] = pool
    with pytest.raises(native.FFXError, match="row 17"):
        native.save_run_coded(tmp_path / "bad.tsv", q_names, pair_off, bad, names, scores, "tag")
    assert not native.save_run_coded(tmp_path / "q.tsv", q_names, pair_off, codes, np.array(['a"b'] * pool, dtype=object), scores, "tag")  # needs a CSV writer


def test_native_unique_rows_equals_numpy_unique():
    """ffx_host_unique_rows (a bitmap over the index's rows, marked and ranked by host threads) against
    numpy.unique(rows, return_inverse=True): same distinct rows in the same order, same inverse; a row outside the
    index is refused."""
    from fast_forward_amd import native

    rng = np.random.default_rng(0)
    for n, n_index in ((1, 10), (1000, 50), (200_000, 150_000), (300_000, 40_000_000), (70_000, 64), (70_000, 65)):
        rows = rng.integers(0, n_index, size=n).astype(np.int64)
        want_u, want_inv = np.unique(rows, return_inverse=True)
        for threads in (0, 1, 5):
            u, inv = native.unique_rows(rows, n_index, threads)
            assert inv.dtype == np.int32 and np.array_equal(u, want_u) and np.array_equal(inv, want_inv), (n, n_index, threads)
    with pytest.raises(native.FFXError, match="outside the index"):
        native.unique_rows(np.array([5, 10], dtype=np.int64), 10)
    with pytest.raises(native.FFXError, match="outside the index"):
        native.unique_rows(np.array([5, -1], dtype=np.int64), 10)
    assert native.unique_rows(np.zeros(0, dtype=np.int64), 10)[0].shape == (0,)


@pytest.mark.parametrize("n_values", [50, 20000])
def test_object_column_numbering_depends_on_the_strings_only(n_values):
    """``native.code_object_column`` numbers a frame column through its object POINTERS first and the distinct strings
    second.  Whatever the objects' addresses and the row order, the name table comes out the same (ranks of a sharded
    index number the same ids alike), ``names[codes]`` gives the column back, and the joined name buffer matches."""
    from fast_forward_amd import native

    rng = np.random.default_rng(3)
    values = np.array([f"doc-{i * 7919 % 100003}" for i in range(n_values)] + ["", "ünï"], dtype=object)
    col = values[rng.integers(0, len(values), size=60000)]
    codes, names, packed = native.code_object_column(col)
    assert codes.dtype == np.int32 and np.array_equal(names[codes], col)
    assert len(set(names.tolist())) == len(names) == len(set(col.tolist()))
    assert packed is not None and packed[1] == len(names) and packed[0].decode("utf-8").split("\n") == names.tolist()
    # fresh objects for every cell, rows shuffled: the same table
    own = np.array([(s + " ")[:-1] for s in col[rng.permutation(len(col))].tolist()], dtype=object)
    codes2, names2, packed2 = native.code_object_column(own)
    assert names2.tolist() == names.tolist() and packed2 == packed and np.array_equal(names2[codes2], own)
    # a cell that is no string: not this function's business
    bad = col.copy()
    bad[17] = 3.5
    assert native.code_object_column(bad) is None
    # a name holding the separator cannot be handed on as a joined buffer
    odd = col.copy()
    odd[5] = "two\nlines"
    assert native.code_object_column(odd)[2] is None
    # without the object array of the names: the same codes and buffer, None where the buffer stands for the table
    codes3, names3, packed3 = native.code_object_column(col, names_as_objects=False)
    assert names3 is None and packed3 == packed and np.array_equal(codes3, codes)
    assert native.code_object_column(odd, names_as_objects=False)[1] is not None  # (no buffer can stand for these names)
