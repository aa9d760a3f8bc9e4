#!/usr/bin/env python3
"""Differential fuzz of the rebuilt run-file reader / writer and ``Ranking.from_columns`` against THE REFERENCE ITSELF.

Runs only in the build container (imports /root/reference with the recipe of oracle/gen_golden.py: SURVEY.md section 8c); test
infrastructure, never shipped, nothing of the reference is copied.  Two parts:

  random     random TREC run files (1 - 400 queries, mixed id shapes, shuffled rows, exact ties, separators tab / spaces, four score
             formats, with and without a final newline): ``Ranking.from_file`` frames equal to the reference's (scores bit for bit,
             order modulo exact ties), equal exception types (repeated pairs), ``save`` equal to the reference's writer line by line
             modulo id swaps inside exact ties, the written file re-read by both, ``from_columns`` (fixed-width and Arrow columns)
             equal to the reference's constructor on the same rows
  awkward    the token shapes pandas.read_csv re-interprets (src/fast_forward/ranking.py:400-409): query / document ids that look
             like integers with leading zeros, floats, exponents, hex, booleans, "NA" / "null" / "nan"; scores "nan", "inf", "1_0",
             ".5", "3.", "+2.5": every (query form x id form x score form) combination, frames or exception types equal

    python oracle/diff_fuzz_ranking_io.py random 0 120      # round 5: 88 runs equal (+ 32 equal errors), seed 0
    python oracle/diff_fuzz_ranking_io.py awkward 1 315     # round 5: 276 equal frames + 39 equal errors: all 315 combinations
"""
import importlib.metadata as _md, sys, types, tempfile, os
from pathlib import Path
import numpy as np, pandas as pd
sys.dont_write_bytecode = True
for _name, _attrs in (("nanopq", ("PQ", "OPQ")), ("h5py", ("File",))):
    _m = types.ModuleType(_name)
    for _a in _attrs: setattr(_m, _a, type(_a, (), {}))
    sys.modules[_name] = _m
_orig = _md.version
_md.version = lambda n: "0.8.0" if n == "fast-forward-indexes" else _orig(n)
sys.path.insert(0, "/root/reference/src")
from fast_forward import Ranking as Ref
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "fast-forward-indexes_amd"))
from fast_forward_amd import Ranking as Mine
import fast_forward_amd.ranking as mr

def frames_equal(a, b, what):
    da, db = a._df.reset_index(drop=True), b._df.reset_index(drop=True)
    assert list(da.columns) == list(db.columns), (what, da.columns, db.columns)
    assert len(da) == len(db), (what, len(da), len(db))
    assert da["q_id"].astype(str).tolist() == db["q_id"].astype(str).tolist(), what
    assert np.array_equal(da["score"].to_numpy().view(np.uint32), db["score"].to_numpy().view(np.uint32)), what
    # ids equal except inside runs of exactly tied scores of one query
    ia, ib = da["id"].astype(str).to_numpy(), db["id"].astype(str).to_numpy()
    diff = np.flatnonzero(ia != ib)
    for i in diff:
        q, s = da["q_id"][i], da["score"][i]
        same = (da["q_id"] == q) & (da["score"] == s)
        assert sorted(ia[same.to_numpy()]) == sorted(ib[same.to_numpy()]), (what, i)


import itertools


def random_runs():
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    work = Path(tempfile.mkdtemp())
    n_ok = 0
    for trial in range(int(sys.argv[3]) if len(sys.argv) > 3 else 120):
        big = trial % 10 == 9
        n_q = int(rng.integers(1, 400 if big else 12))
        rows = []
        for q in range(n_q):
            qid = rng.choice([f"q{q}", f"{q}", f"Q-{q:03d}", f"q{q}.x"])
            k = int(rng.integers(1, 300 if big else 30))
            docs = rng.choice(5000, size=k, replace=False)
            for r, dno in enumerate(docs):
                did = rng.choice([f"D{dno}", f"{dno}", f"doc_{dno}#p{r % 3}"])
                sc = rng.choice([float(rng.normal() * 10), float(int(rng.integers(-5, 5))), 1e-7 * float(rng.normal()), 12345.678])
                rows.append((str(qid), str(did), sc, r))
        # possibly shuffled rows, ties, a tag
        if rng.random() < 0.5:
            rng.shuffle(rows)
        tag = rng.choice(["bm25", "run-1", "x"])
        sep = rng.choice(["\t", " ", "  ", " \t"])
        fmt = rng.choice(["%r", "%.6f", "%.3g", "%d"])
        lines = []
        for qid, did, sc, r in rows:
            if fmt == "%d": sc_s = str(int(sc))
            elif fmt == "%r": sc_s = repr(sc)
            else: sc_s = fmt % sc
            lines.append(sep.join([qid, "Q0", did, str(r), sc_s, tag]))
        text = "\n".join(lines) + ("\n" if rng.random() < 0.8 else "")
        p = work / f"run_{trial}.tsv"
        p.write_text(text)
        # duplicate pairs -> both must raise ValueError
        try:
            ref = Ref.from_file(p)
            ref_err = None
        except Exception as e:
            ref, ref_err = None, type(e)
        try:
            mine = Mine.from_file(p)
            mine_err = None
        except Exception as e:
            mine, mine_err = None, type(e)
        assert ref_err == mine_err, (trial, ref_err, mine_err)
        if ref is None:
            continue
        frames_equal(mine, ref, f"from_file {trial}")
        assert mine.name == ref.name, (mine.name, ref.name)
        # save: same bytes as the reference's writer unless tie order differs
        ref.save(work / "ref_out.tsv"); mine.save(work / "mine_out.tsv")
        a, b = (work / "ref_out.tsv").read_text().splitlines(), (work / "mine_out.tsv").read_text().splitlines()
        assert len(a) == len(b)
        bad = [i for i, (x, y) in enumerate(zip(a, b)) if x != y]
        for i in bad:  # only id swaps inside exact ties
            xa, xb = a[i].split("\t"), b[i].split("\t")
            assert xa[0] == xb[0] and xa[3:] == xb[3:], (trial, a[i], b[i])
        # re-read what I wrote with the reference
        frames_equal(Mine.from_file(work / "mine_out.tsv"), Ref.from_file(work / "mine_out.tsv"), f"reread {trial}")
        # from_columns vs the reference constructor on the same rows
        df = pd.DataFrame({"q_id": [r[0] for r in rows], "id": [r[1] for r in rows], "score": np.array([r[2] for r in rows], dtype=np.float32)})
        ref2 = Ref(df.copy())
        for form in ("S", "arrow"):
            if form == "S":
                qcol, icol = df["q_id"].to_numpy().astype("S"), df["id"].to_numpy().astype("S")
            else:
                import pyarrow as pa
                qcol, icol = pa.array(df["q_id"].to_numpy(), type=pa.string()), pa.array(df["id"].to_numpy(), type=pa.string())
            m2 = Mine.from_columns(qcol, icol, df["score"].to_numpy())
            frames_equal(m2, ref2, f"from_columns {form} {trial}")
        n_ok += 1
    print("differential fuzz ok:", n_ok, "runs compared with the reference")


def awkward_tokens():

    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    work = Path(tempfile.mkdtemp())
    qid_forms = [lambda q: f"q{q}", lambda q: f"{q}", lambda q: f"{q:03d}", lambda q: f"{q}.0", lambda q: f"1e{q%3}", lambda q: "NA" if q == 0 else f"q{q}", lambda q: f"0x{q:x}"]
    did_forms = [lambda d: f"D{d}", lambda d: f"{d}", lambda d: f"{d:07d}", lambda d: f"{d}.5", lambda d: "null" if d % 97 == 0 else f"D{d}", lambda d: f"{d}e2", lambda d: "nan" if d % 89 == 0 else f"{d}", lambda d: f"-{d}", lambda d: "True" if d % 2 else "False"]
    score_forms = ["%r", "%.4f", "%d", "%.3e", "special"]
    n_ok = n_err = 0
    cases = list(itertools.product(range(len(qid_forms)), range(len(did_forms)), range(len(score_forms))))
    rng.shuffle(cases)
    for trial, (qi, di, si) in enumerate(cases[: int(sys.argv[3]) if len(sys.argv) > 3 else 315]):
        n_q = int(rng.integers(1, 6))
        lines = []
        for q in range(n_q):
            k = int(rng.integers(1, 12))
            for r, dno in enumerate(rng.choice(500, size=k, replace=False)):
                sc = float(rng.normal() * 10)
                if score_forms[si] == "special":
                    sc_s = str(rng.choice(["nan", "NaN", "inf", "-inf", "1e3", "5", "-0.0", "+2.5", "1_0", ".5", "3."]))
                elif score_forms[si] == "%d":
                    sc_s = str(int(sc))
                else:
                    sc_s = score_forms[si] % sc if score_forms[si] != "%r" else repr(sc)
                lines.append("\t".join([qid_forms[qi](q), "Q0", did_forms[di](int(dno)), str(r), sc_s, "tag"]))
        p = work / f"case_{trial}.tsv"
        p.write_text("\n".join(lines) + "\n")
        try:
            ref = Ref.from_file(p); ref_err = None
        except Exception as e:
            ref, ref_err = None, type(e).__name__
        try:
            mine = Mine.from_file(p); mine_err = None
        except Exception as e:
            mine, mine_err = None, type(e).__name__
        if ref_err != mine_err:
            print("ERROR MISMATCH", trial, (qi, di, si), ref_err, mine_err, p); sys.exit(1)
        if ref is None:
            n_err += 1; continue
        try:
            frames_equal(mine, ref, f"case {trial} {(qi, di, si)}")
        except AssertionError as e:
            print("FRAME MISMATCH", (qi, di, si), p); print(ref._df.head(8)); print(mine._df.head(8)); sys.exit(1)
        n_ok += 1
    print("awkward-token fuzz ok:", n_ok, "equal frames,", n_err, "equal errors")


if __name__ == "__main__":
    (awkward_tokens if len(sys.argv) > 1 and sys.argv[1] == "awkward" else random_runs)()
