#!/opt/conda/bin/python3.9
"""Write small HDF5 index files with the layout the reference's OnDiskIndex produces.

The reference cannot be imported under this interpreter (it needs Python >= 3.10) and h5py does
not exist for the main interpreter, so this script REPLAYS the h5py calls of
src/fast_forward/index/disk.py (file attrs :83-85, datasets :145-165, quantizer groups :123-136,
writes :243-301) with the only h5py available here (Anaconda Python 3.9, h5py 3.3.0 / HDF5 1.10.6).
The files are data: tests/golden/h5/*.h5 plus a JSON with the expected content, used to check the
build's own HDF5 reader (fast_forward_amd/index/h5read.py).

    /opt/conda/bin/python3.9 oracle/gen_h5_fixture.py
"""
import json
from pathlib import Path

import h5py
import numpy as np

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "h5"


def create(path, dim, dtype, init_size, chunk_size, max_id_length):
    fp = h5py.File(path, "w")
    fp.attrs["num_vectors"] = 0
    fp.attrs["ff_version"] = "0.8.0"
    return fp


def create_ds(fp, dim, dtype, init_size, chunk_size, max_id_length):
    fp.create_dataset("vectors", (init_size, dim), dtype, maxshape=(None, dim), chunks=(chunk_size, dim))
    fp.create_dataset("doc_ids", (init_size,), f"S{max_id_length}", maxshape=(None,), chunks=True)
    fp.create_dataset("psg_ids", (init_size,), f"S{max_id_length}", maxshape=(None,), chunks=True)


def add(fp, vectors, doc_ids, psg_ids, chunk_size):
    n0 = int(fp.attrs["num_vectors"])
    n_new = len(vectors)
    cap = fp["vectors"].shape[0]
    if n_new > cap - n0:
        new_size = int((n0 + n_new) / chunk_size + 0.5) * chunk_size
        for ds in ("vectors", "doc_ids", "psg_ids"):
            fp[ds].resize(new_size, axis=0)
    di = [n0 + i for i, d in enumerate(doc_ids) if d is not None]
    if di:
        fp["doc_ids"][di] = [d for d in doc_ids if d is not None]
    pi = [n0 + i for i, p in enumerate(psg_ids) if p is not None]
    if pi:
        fp["psg_ids"][pi] = [p for p in psg_ids if p is not None]
    fp["vectors"][n0:n0 + n_new] = vectors
    fp.attrs["num_vectors"] += n_new


def store_quantizer(fp, meta, attributes, data):
    fp.create_group("quantizer/meta").attrs.update(meta)
    fp.create_group("quantizer/attributes").attrs.update(attributes)
    g = fp.create_group("quantizer/data")
    for k, v in data.items():
        g.create_dataset(k, data=v)


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    rng = np.random.default_rng(0)
    expected = {}

    # 1. float32 vectors, doc+psg ids, several chunks, resized twice
    n, dim, cs = 23, 16, 4
    vec = rng.standard_normal((n, dim)).astype(np.float32)
    doc_ids = [f"d{i // 3}" if i % 5 else None for i in range(n)]
    psg_ids = [f"p{i}" if i % 7 else None for i in range(n)]
    for i in range(n):
        if doc_ids[i] is None and psg_ids[i] is None:
            psg_ids[i] = f"p{i}"
    fp = create(OUT / "plain.h5", dim, np.float32, 8, cs, 8)
    create_ds(fp, dim, np.float32, 8, cs, 8)
    add(fp, vec[:5], doc_ids[:5], psg_ids[:5], cs)
    add(fp, vec[5:14], doc_ids[5:14], psg_ids[5:14], cs)
    add(fp, vec[14:], doc_ids[14:], psg_ids[14:], cs)
    fp.close()
    expected["plain.h5"] = {"num_vectors": n, "dim": dim, "dtype": "float32", "vectors": vec.tolist(), "doc_ids": doc_ids, "psg_ids": psg_ids,
                            "ff_version": "0.8.0"}

    # 2. a bigger one: default-ish chunking (chunk 1024 rows), 3000 vectors of dim 64, float64 as the
    #    reference stores what it is given (np.random.normal is float64 in its tests)
    n, dim, cs = 1400, 32, 512
    vec = rng.standard_normal((n, dim))
    doc_ids = [f"D{i // 2}" for i in range(n)]
    psg_ids = [f"P{i}" for i in range(n)]
    fp = create(OUT / "big_f64.h5", dim, np.float64, 1024, cs, 12)
    create_ds(fp, dim, np.float64, 1024, cs, 12)
    add(fp, vec[:700], doc_ids[:700], psg_ids[:700], cs)
    add(fp, vec[700:], doc_ids[700:], psg_ids[700:], cs)
    fp.close()
    sample = [0, 1, 699, 700, 1399]
    expected["big_f64.h5"] = {"num_vectors": n, "dim": dim, "dtype": "float64", "sample_rows": sample,
                              "vectors_sample": vec[sample].tolist(), "n_docs": n // 2, "ff_version": "0.8.0",
                              "row_sums": vec.sum(axis=1).tolist()}

    # 3. quantized: uint8 codes + quantizer groups (meta / attributes attrs, data/codewords dataset)
    M, Ks, Ds, n = 4, 8, 6, 11
    codewords = rng.standard_normal((M, Ks, Ds)).astype(np.float32)
    codes = rng.integers(0, Ks, size=(n, M)).astype(np.uint8)
    doc_ids = [f"q{i}" for i in range(n)]
    fp = create(OUT / "quantized.h5", M, np.uint8, 16, 8, 8)
    store_quantizer(fp, {"__module__": "fast_forward.quantizer.nanopq", "__name__": "NanoPQ", "_trained": True},
                    {"M": M, "Ks": Ks, "Ds": Ds, "metric": "dot", "verbose": False}, {"codewords": codewords})
    create_ds(fp, M, np.uint8, 16, 8, 8)
    add(fp, codes, doc_ids, [None] * n, 8)
    fp.close()
    expected["quantized.h5"] = {"num_vectors": n, "dim": M, "dtype": "uint8", "vectors": codes.tolist(), "doc_ids": doc_ids,
                                "psg_ids": [None] * n, "codewords": codewords.tolist(), "ff_version": "0.8.0",
                                "meta": {"__module__": "fast_forward.quantizer.nanopq", "__name__": "NanoPQ", "_trained": True},
                                "attributes": {"M": M, "Ks": Ks, "Ds": Ds, "metric": "dot", "verbose": False}}

    # 4. empty index (attrs only)
    fp = create(OUT / "empty.h5", 0, np.float32, 0, 0, 8)
    fp.close()
    expected["empty.h5"] = {"num_vectors": 0}

    (OUT / "expected.json").write_text(json.dumps(expected))
    for f in sorted(OUT.iterdir()):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
