#!/usr/bin/env python3
"""Headline benchmark: re-ranked queries/sec at k=5000, 768-d fp32, MS MARCO passage scale.

    python bench.py [--gpus N --steps K --warmup W]

N > 1: one rank per GPU.  Either launched by `python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or plainly as
`python bench.py --gpus N`: then this process starts the N ranks itself (before it touches the GPU),
relays rank 0's JSON line and exits with their status.  The ranks talk through libffx's own
communicator (RCCL over xGMI; `--transport host` = shared memory, for ranks that share a GPU).

One *step* = one pass of the re-ranking hot path over one batch of synthetic first-stage
candidates: expand candidates -> gather_dot (HIP) -> MaxP segment reduce + alpha interpolation ->
per-query ordering (N > 1: local gather_dot -> all-to-all of scores -> owner-side interpolation +
ordering).  All inputs (index, query vectors, candidate CSR, sparse scores) are resident in HBM
before the timed region; outputs stay in HBM.

Prints ONE JSON line (rank 0) with the fields the driver expects plus
  "roofline":     achieved algorithmic GB/s of the gather_dot kernel vs the 8 TB/s HBM peak,
                  kernel time measured with HIP events on the kernel's own stream
  "cpu_baseline": the reference-equivalent NumPy path (oracle port) timed on this host on a
                  bounded sample of the same workload
  "regions":      SURVEY.md 8(d)'s timed regions side by side: (i) device-resident without / with the
                  ordering, (ii) host buffers in -> host buffers out through ffx_score (page-locked and
                  pageable), (iii) Index.rerank from a Ranking of strings; and what the box streams.
"""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
import uuid
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "fast-forward-indexes_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from fast_forward_amd import native, synth  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 TB/s achievable copy)
MODE_NO = {"PASSAGE": 1, "MAXP": 2, "FIRSTP": 3, "AVEP": 4}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=8_841_823, help="index rows (MS MARCO passage count)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", type=int, default=6980)
    ap.add_argument("--k", type=int, default=5000, help="candidates per query")
    ap.add_argument("--mode", default="MAXP", choices=list(MODE_NO))
    ap.add_argument("--layout", default="single", choices=["single", "multi"],
                    help="single: 1 passage per document (MS MARCO passage); multi: 1..8 passages per document")
    ap.add_argument("--sum", default="exact", choices=["exact", "fast"], help="summation order of gather_dot")
    ap.add_argument("--store", default="f32", choices=["f32", "f16", "pq"], help="row storage: float32, float16, or PQ codes (M=8, Ks=256)")
    ap.add_argument("--alpha", type=float, default=0.1)
    ap.add_argument("--sort", dest="sort", action="store_true", default=True,
                    help="order every query's candidates on the device as part of the step (default)")
    ap.add_argument("--no-sort", dest="sort", action="store_false", help="leave the ordering out of the step")
    ap.add_argument("--placement-tries", type=int, default=-1,
                    help="ffx_index_reseat after the index is filled (what InMemoryIndex.consolidate() does): the rows move into the fastest of "
                         "up to this many further allocations (-1, the library's default: as many as half of the free memory holds, at most 7); "
                         "0: the allocation stays where hipMalloc put it")
    ap.add_argument("--no-regions", dest="regions", action="store_false", default=True,
                    help="skip the side measurements (host-to-host, Index API, stream ceiling) of the 1-GPU line")
    ap.add_argument("--api-queries", type=int, default=6980, help="queries of the Index-API region (iii); 0: skip")
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--cpu-sample-queries", type=int, default=64, help="queries in the CPU-baseline sample (0: skip)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work to measure for the baseline")
    ap.add_argument("--traffic-bytes", type=float, default=None, help="HBM bytes per gather_dot launch from a rocprofv3 --pmc pass")
    ap.add_argument("--no-traffic", dest="traffic", action="store_false", default=True,
                    help="do not run the rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) that fill roofline.traffic in the 1-GPU line")
    ap.add_argument("--force-sharded", action="store_true", help="run the multi-GPU leg even with one rank (rehearsal)")
    ap.add_argument("--sharding", default="rows", choices=["rows", "queries"],
                    help="multi-GPU layout: rows = index row-sharded by document + all-to-all of scores (north star); "
                         "queries = every rank holds the whole index and scores its block of the queries, no exchange "
                         "(the zero-communication comparison point when the index fits one GPU)")
    ap.add_argument("--transport", default=None, choices=["rccl", "host"],
                    help="rccl (default): RCCL over xGMI, one GPU per rank; host: the exchange staged through shared "
                         "memory, so N ranks can rehearse on ONE GPU (RCCL refuses ranks that share a device)")
    ap.add_argument("--dist-backend", default=None, choices=["nccl", "gloo"], help="older spelling of --transport (gloo = host)")
    ap.add_argument("--fresh-queries", default="allgather", choices=["allgather", "replicate"],
                    help="fresh batches of the sharded leg: every rank uploads the vectors of the queries it OWNS and the ranks all-gather them on the "
                         "routing lane (intake per rank ~ 1/N), or every rank uploads ALL query vectors over its own PCIe link (no exchange)")
    ap.add_argument("--nccl-channels", default=None, metavar="MIN:MAX",
                    help="N > 1 over RCCL: export NCCL_MIN_NCHANNELS / NCCL_MAX_NCHANNELS before the communicator comes up (how many workgroups RCCL's "
                         "send/recv kernel may take beside gather_dot: the first knob if a measured scaling point falls short of the rehearsal); echoed in the line")
    ap.add_argument("--rehearsal-exchange", default="copy", metavar="copy|paced[:GB/s]",
                    help="--as-rank-of N: what stands in for the peers' traffic -- `copy`: one kernel of 16 workgroups moving the bytes through this GPU's memory; "
                         "`paced[:R]`: one PERSISTENT workgroup per peer and direction that moves its piece in 64 KiB chunks and spins on a flag between chunks "
                         "until a link of R GB/s (default 50) would have delivered them")
    ap.add_argument("--fresh-timeout", type=float, default=300.0, help="seconds the fresh-batch section of the sharded leg may take before the line is printed without it")
    ap.add_argument("--require-rccl", action="store_true",
                    help="N > 1: exit non-zero if RCCL cannot be brought up on every rank instead of falling back to the shared-memory "
                         "transport (a fallback line is marked `\"scaling\": \"invalid: host transport\"` and `\"value_is_rccl\": false`)")
    ap.add_argument("--as-rank-of", type=int, default=0, metavar="N",
                    help="single-GPU REHEARSAL of one rank of an N-rank row-sharded run: this process holds rank --as-rank's shard "
                         "(rows/N by doc_partition), scores its local share (~P/N pairs), owns ~Q/N queries, and the exchange moves the "
                         "same bytes through this GPU's memory (peers do not exist; their scores arrive as zeros).  Times one rank's "
                         "step; the line is marked as a rehearsal and is not a scaling measurement")
    ap.add_argument("--as-rank", type=int, default=0, metavar="R", help="which rank --as-rank-of N rehearses (default 0)")
    args = ap.parse_args()
    if args.transport is None:
        args.transport = "host" if args.dist_backend == "gloo" else "rccl"
    if args.as_rank_of:
        if args.as_rank_of < 1 or not 0 <= args.as_rank < args.as_rank_of:
            ap.error("--as-rank-of N needs N >= 1 and 0 <= --as-rank < N")
        if args.gpus != 1:
            ap.error("--as-rank-of is a single-process rehearsal: leave --gpus at 1")
        args.transport = "rehearsal"
    return args


def metric_name(args) -> str:
    store = {"f32": "fp32", "f16": "fp16", "pq": "PQ(M=8,Ks=256)"}[args.store]
    return f"re-ranked queries/sec at k={args.k}, {args.dim}-d {store}; achieved HBM GB/s vs peak"


def compute_dtype(args) -> str:
    """The arithmetic type of the path: fp32 products and sums whatever the rows are stored as."""
    return {"f32": "f32", "f16": "f32 (rows stored as f16, widened exactly)", "pq": "f32 (rows stored as u8 PQ codes)"}[args.store]


def doc_layout(n_rows: int, layout: str, seed: int):
    """doc ordinal -> contiguous row range.  Returns (doc_off int64[D+1])."""
    if layout == "single":
        return np.arange(n_rows + 1, dtype=np.int64)
    # passages per document ~ U{1..8}; deterministic from the seed, trimmed to n_rows
    n_guess = int(n_rows / 4.5) + 16
    per = (synth.uniform_u32(np.arange(n_guess, dtype=np.uint64), seed + 7) % np.uint32(8)).astype(np.int64) + 1
    off = np.concatenate([[0], np.cumsum(per)])
    n_docs = int(np.searchsorted(off, n_rows, side="right")) - 1
    off = off[: n_docs + 1].copy()
    if off[-1] < n_rows:
        off = np.append(off, n_rows)
    return off.astype(np.int64)


def candidates(n_docs: int, n_queries: int, k: int, seed: int, q_first: int = 0):
    """k distinct documents per query, uniform over the collection: candidate i of a query is
    drawn uniformly from stratum i of k equal strata (distinct by construction)."""
    stratum = n_docs // k
    assert stratum >= 1, "k larger than the number of documents"
    idx = (np.arange(n_queries * k, dtype=np.uint64) + np.uint64(q_first * k))
    r = synth.uniform_u32(idx, seed + 1) % np.uint32(stratum)
    # visit the strata in a per-query scrambled order (i -> (A*i + B_q) mod k, A coprime to k):
    # a first-stage run is ordered by score, not by document id
    a = next(c for c in range(max(1, int(0.618 * k)), 4 * k + 7) if np.gcd(c, k) == 1)  # golden-ratio stride
    b = synth.uniform_u32(np.arange(q_first, q_first + n_queries, dtype=np.uint64), seed + 5).astype(np.int64) % k
    strata = (np.arange(k, dtype=np.int64)[None, :] * a + b[:, None]) % k
    ids = (strata.reshape(-1) * stratum + r.astype(np.int64)).astype(np.int32)
    sparse = (synth.uniform_u32(idx, seed + 2).astype(np.float32) * np.float32(25.0 / 2**32)).astype(np.float32)
    return ids, sparse


def usable_cores() -> int:
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota
    (a container on a 256-core host is often allowed 16; 256 OpenMP threads on that quota crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), None))):
        try:
            quota, period = parse(open(path).read())
            if period is None:
                period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def cpu_baseline(args, doc_off, n_docs, scale, q_first: int = 0, batch_seed: int | None = None):
    """Reference-equivalent CPU path (oracle/ffx_oracle.py numpy_port_score: the reference's
    own three NumPy ops, index/base.py:301-303) on the first `cpu_sample_queries` queries of
    the same workload.  The rows those queries touch are re-derived on the host with the same
    counter-based generator the device index was filled with."""
    sys.path.insert(0, str(ROOT / "oracle"))
    os.environ.setdefault("OMP_NUM_THREADS", str(usable_cores()))  # read by libgomp when the oracle library loads
    import ffx_oracle as fo

    nq = args.cpu_sample_queries
    ids, sparse = candidates(n_docs, nq, args.k, args.seed if batch_seed is None else batch_seed, q_first=q_first)
    pair_off = np.arange(nq + 1, dtype=np.int64) * args.k
    # compact the touched documents into a small host index (same row values, renumbered)
    uniq, inv = np.unique(ids, return_inverse=True)
    counts = (doc_off[uniq + 1] - doc_off[uniq]).astype(np.int64)
    sub_off = np.concatenate([[0], np.cumsum(counts)])
    rows_global = np.concatenate([np.arange(doc_off[u], doc_off[u + 1]) for u in uniq]) if args.layout == "multi" else doc_off[uniq]
    if args.store == "pq":  # rebuild the decoded rows the device scores: codes -> codewords
        m, ks, ds = 8, 256, args.dim // 8
        book = synth.rows(args.seed + 9, scale, 0, m * ks, ds).reshape(m, ks, ds)
        elem = (rows_global[:, None] * m + np.arange(m)[None, :]).astype(np.uint64)
        codes = ((synth.mix64(elem.reshape(-1), args.seed) >> np.uint64(32)) % np.uint64(ks)).astype(np.int64).reshape(-1, m)
        vecs = np.concatenate([book[j][codes[:, j]] for j in range(m)], axis=1)
    else:
        vecs = fo.synth_row_list(args.seed, scale, rows_global, args.dim)
    if args.store == "f16":  # the index rounds rows to float16 once; NumPy then widens them back
        vecs = vecs.astype(np.float16).astype(np.float32)
    queries = synth.rows(args.seed + 3, scale, q_first, nq, args.dim)
    sub_rows = np.arange(rows_global.shape[0], dtype=np.int32)
    mode = MODE_NO[args.mode]
    # repeat whole passes over the sample until ~10 s of CPU work has been measured
    t_np, passes = 0.0, 0
    while t_np < args.cpu_seconds and passes < 1000:
        t0 = time.perf_counter()
        ff, interp = fo.numpy_port_score(vecs, sub_off, sub_rows, sub_rows, queries, pair_off, inv.astype(np.int32), mode,
                                         sparse, args.alpha, batch_queries=4)
        t_np += time.perf_counter() - t0
        passes += 1
    t_c, passes_c = 0.0, 0
    while t_c < args.cpu_seconds / 4 and passes_c < 1000:
        t0 = time.perf_counter()
        ff_c, interp_c = fo.score(vecs, sub_off, sub_rows, sub_rows, queries, pair_off, inv.astype(np.int32), mode, sparse, args.alpha)
        t_c += time.perf_counter() - t0
        passes_c += 1
    assert np.array_equal(ff.view(np.uint32), ff_c.view(np.uint32))
    # ... and the same sample the way the reference's users run it: frames of string ids, a dictionary look-up per distinct id,
    # merges + group-by around the three NumPy lines, an outer merge and two sorts (SURVEY 8(d)'s "pandas-faithful variant")
    t_pd, passes_pd, pd_steps = 0.0, 0, {}
    while t_pd < args.cpu_seconds / 4 and passes_pd < 1000:
        ff_p, interp_p, pd_steps = fo.pandas_port_score(vecs, sub_off, sub_rows, sub_rows, queries, pair_off, inv.astype(np.int32), mode, sparse, args.alpha,
                                                        batch_queries=16)
        t_pd += pd_steps["total_s"]
        passes_pd += 1
    assert np.array_equal(ff_p.view(np.uint32), ff.view(np.uint32)) and np.array_equal(interp_p.view(np.uint32), interp.view(np.uint32))
    threads = int(os.environ["OMP_NUM_THREADS"])
    out = {
        "value": nq * passes / t_np, "unit": "queries/s", "cores": 1, "kind": "port",
        "sample": f"first {nq} queries x k={args.k} ({nq * args.k} pairs, {rows_global.shape[0]} distinct rows = "
                  f"{vecs.nbytes / 1e6:.0f} MB re-derived on the host) of the same workload, {passes} whole passes "
                  f"({t_np:.1f} s) of the NumPy port of index/base.py:301-303 + ranking.py:319, 4 queries per batch; "
                  f"host has {os.cpu_count()} cores ({usable_cores()} usable by this process), NumPy elementwise ops use 1",
        "seconds": t_np, "passes": passes,
        "omp_c_port": {"value": nq * passes_c / t_c, "unit": "queries/s", "cores": threads, "seconds": t_c, "passes": passes_c,
                       "note": "oracle/ffx_oracle.c (scalar, NumPy summation order), OpenMP over candidates"},
        "pandas_port": {"value": nq * passes_pd / t_pd, "unit": "queries/s", "cores": 1, "seconds": t_pd, "passes": passes_pd,
                        "steps_of_last_pass_s": {k: round(v, 4) for k, v in pd_steps.items()},
                        "note": "oracle/ffx_oracle.py pandas_port_score: index(first) then first.interpolate(dense, alpha) through frames of string ids "
                                "(index/base.py:389-469 with batch_size = 16 queries, :279-314, index/util.py:12-42, ranking.py:67-121, :293-326), same sample, "
                                "same bits; the frames and id dictionaries exist before the clock starts"},
    }
    return out, (ids, sparse, interp)


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes BEFORE this
    process makes any GPU call (it never does), let them print (rank 0 prints the JSON line), and
    return their status.  Never exec from a process that has touched the GPU."""
    key = f"bench_{os.getpid()}_{uuid.uuid4().hex[:12]}"
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   FFX_RDZV_KEY=key, MASTER_ADDR="127.0.0.1")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env))
    status, deadline = 0, None
    while any(p.poll() is None for p in procs):
        for p in procs:
            if p.poll() not in (None, 0) and status == 0:
                status = p.returncode  # one rank failed: give the others a moment (they fail at their next collective), then stop them
                deadline = time.monotonic() + 30
        if deadline is not None and time.monotonic() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()  # the exact children started above
        time.sleep(0.05)
    for p in procs:
        if p.returncode != 0 and status == 0:
            status = p.returncode
    return status


def gather_stats(comm, st: dict) -> list[dict]:
    """Every rank's per-step figures, in rank order (control plane of the communicator)."""
    keys = ("ms_gather_dot", "ms_expand", "ms_total", "n_rows", "n_pairs", "own_pairs", "own_queries", "send_bytes", "recv_bytes", "kernel_variant",
            "ms_exchange", "ms_merge")
    mine = np.array([float(st[k]) for k in keys], dtype=np.float64)
    return [dict(zip(keys, row.tolist())) for row in comm.allgather_array(mine)]


def apply_comm_knobs(args) -> dict:
    """Environment the communicator reads when it comes up; returns what was set (echoed in the line)."""
    knobs = {}
    if args.nccl_channels:
        lo, _, hi = args.nccl_channels.partition(":")
        os.environ["NCCL_MIN_NCHANNELS"], os.environ["NCCL_MAX_NCHANNELS"] = lo, (hi or lo)
    for name in ("NCCL_MIN_NCHANNELS", "NCCL_MAX_NCHANNELS", "NCCL_NCHANNELS_PER_PEER", "NCCL_P2P_NET_CHUNKSIZE", "NCCL_PROTO", "GPU_MAX_HW_QUEUES"):
        if name in os.environ:
            knobs[name] = os.environ[name]
    if args.as_rank_of and args.rehearsal_exchange.startswith("paced"):
        _, _, rate = args.rehearsal_exchange.partition(":")
        word = "rehearsal_paced" + (f"={int(float(rate))}" if rate else "")
        os.environ["FFX_DEBUG_VARIANT"] = ",".join(w for w in (os.environ.get("FFX_DEBUG_VARIANT"), word) if w)
    return knobs


def bench_sharded(args, rank: int, world: int):
    """bench.py --gpus N>1: the SAME workload as N=1 (fixed index, queries and k: strong
    scaling), index row-sharded by document, one rank per GPU, scores exchanged through libffx's
    communicator (RCCL, or shared memory with --transport host)."""
    from fast_forward_amd.sharded import ShardedScorer, connect, doc_partition, query_partition

    args.comm_knobs = apply_comm_knobs(args)
    if args.as_rank_of:  # one process standing in for one rank: no peers, no rendezvous
        native.require_device()
        comm = connect("rehearsal", rank=args.as_rank, world=args.as_rank_of, device=0)
        return _bench_sharded_body(args, comm.rank, comm.world, comm, comm.device, doc_partition, query_partition, ShardedScorer)
    visible = native.device_count()
    masked = any(os.environ.get(v) for v in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    device = None
    if args.transport == "rccl" and visible < world:
        if not (masked and visible >= 1):
            sys.exit(f"--gpus {world} over RCCL needs {world} GPUs, {visible} visible (use --transport host to rehearse on fewer)")
        device = int(os.environ.get("LOCAL_RANK", str(rank))) % visible  # a launcher gave every rank its own device mask
    # RCCL prints a version banner to STDOUT when the first communicator comes up; stdout of this program is the
    # one JSON line, so anything the libraries print while connecting goes to stderr
    sys.stdout.flush()
    keep_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        comm = _connect_with_fallback(args, rank, world, device, connect)
    finally:
        sys.stdout.flush()
        os.dup2(keep_stdout, 1)
        os.close(keep_stdout)
    local_rank = comm.device
    # a second, GPU-less channel over shared memory that the ranks only use to tell each other how their fresh-batch section
    # went before the closing collective (a rank that failed there must not leave its peers waiting in an all-reduce)
    args.votes = None
    if world > 1:
        from fast_forward_amd.sharded import _rendezvous_key

        try:
            args.votes = connect("host", rank=rank, world=world, device=-1, key=_rendezvous_key() + "votes", timeout=600.0)
        except (native.FFXError, TimeoutError, OSError) as e:
            print(f"rank {rank}: no side channel for the closing vote ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
    return _bench_sharded_body(args, rank, world, comm, local_rank, doc_partition, query_partition, ShardedScorer)


def _guarded(fn, seconds: float, what: str):
    """fn() on a daemon thread; TimeoutError if it has not returned after `seconds` (a collective that
    never completes cannot be cancelled, but the rank can stop waiting for it)."""
    import threading

    box = {}

    def run():
        try:
            box["value"] = fn()
        except BaseException as e:  # noqa: BLE001 - handed to the caller
            box["error"] = e

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(seconds)
    if t.is_alive():
        raise TimeoutError(f"{what} did not finish within {seconds:.0f} s")
    if "error" in box:
        raise box["error"]
    return box["value"]


def _connect_with_fallback(args, rank: int, world: int, device, connect):
    """The job's communicator.  --transport rccl on several ranks: RCCL is brought up and PROBED (barrier,
    all-reduce, one all-to-all through the send/recv mesh, result checked) under a watchdog, and the ranks
    then vote over a shared-memory side channel, so that either every rank continues on RCCL or every rank
    falls back to the shared-memory transport (the line's `transport` says which and why).  A scaling run
    never hangs on, or dies of, a fabric that will not come up on this node."""
    if args.transport != "rccl" or world == 1:
        return connect(args.transport, rank=rank, world=world, device=device)
    os.environ["FFX_COMM_TIMEOUT_S"] = "900"  # ranks may reach the vote minutes apart
    visible = max(1, native.device_count())
    side = connect("host", rank=rank, world=world, device=(rank if device is None else device) % visible, timeout=600.0)
    why = ""

    def bring_up():
        comm = connect("rccl", rank=rank, world=world, device=device, timeout=300.0)
        comm.barrier()
        total = comm.allreduce([float(rank + 1)])[0]
        if total != world * (world + 1) / 2:
            raise RuntimeError(f"all-reduce over RCCL returned {total}")
        # rank r sends (d + 1) values "1000 r + d" to every rank d and checks what the others sent to it
        send = np.concatenate([np.full(d + 1, 1000.0 * rank + d, dtype=np.float32) for d in range(world)])
        got = comm.exchange_host(send, [d + 1 for d in range(world)], [rank + 1] * world)
        want = np.concatenate([np.full(rank + 1, 1000.0 * r + rank, dtype=np.float32) for r in range(world)])
        if not np.array_equal(got, want):
            raise RuntimeError("the all-to-all probe over RCCL returned wrong values")
        return comm

    comm = None
    try:
        comm = _guarded(bring_up, 420.0, "bringing RCCL up")
    except (native.FFXError, TimeoutError, RuntimeError, OSError) as e:
        why = f"{type(e).__name__}: {e}"
        print(f"rank {rank}: RCCL unavailable ({why})", file=sys.stderr, flush=True)
    votes = side.allgather_array(np.array([1.0 if comm is not None else 0.0]))
    if votes.min() > 0:
        side.close()
        return comm
    if args.require_rccl:
        # every rank leaves here (the vote was collective), none re-executes itself: the launcher sees a non-zero status
        print(f"rank {rank}: --require-rccl and {int(votes.shape[0] - votes.sum())} of {world} ranks could not use RCCL ({why or 'this rank could'})",
              file=sys.stderr, flush=True)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(3)  # a communicator that never came up may hold a thread inside RCCL: do not wait for it
    reasons = [r.rstrip(b"\0").decode(errors="replace") for r in side.allgather_bytes(why.encode()[:160].ljust(160, b"\0"))]
    first = next((f"rank {r}: {t}" for r, t in enumerate(reasons) if t), "")
    print(f"rank {rank}: {int(votes.shape[0] - votes.sum())} of {world} ranks could not use RCCL; every rank falls back to the shared-memory transport",
          file=sys.stderr, flush=True)
    args.transport = f"host (rccl unavailable: {first[:200]})"
    args.abandoned_comm = comm  # not closed: destroying a communicator whose peers never arrived can block
    return side


def _bench_sharded_body(args, rank, world, comm, local_rank, doc_partition, query_partition, ShardedScorer):
    if args.sharding == "queries":
        return bench_replicas(args, comm)
    scale = float(1.0 / np.sqrt(args.dim))
    doc_off = doc_layout(args.rows, args.layout, args.seed)
    n_docs = doc_off.shape[0] - 1
    doc_bounds = doc_partition(doc_off, world)
    d_lo, d_hi = int(doc_bounds[rank]), int(doc_bounds[rank + 1])
    r_lo, r_hi = int(doc_off[d_lo]), int(doc_off[d_hi])
    n_local = r_hi - r_lo

    store = np.float16 if args.store == "f16" else np.float32
    if args.store == "pq":
        sys.exit("--store pq is a single-GPU leg")
    shard = native.DeviceIndex(max(n_local, 1), args.dim, device=local_rank, dtype=store)
    shard.fill_synthetic(args.seed, scale, 0, n_local, first_global_row=r_lo)
    placement = shard.reseat(args.placement_tries)
    shard.set_doc_rows(doc_off[d_lo:d_hi + 1] - r_lo, np.arange(n_local, dtype=np.int32))
    shard.set_psg_rows(np.arange(n_local, dtype=np.int32))
    ids, sparse = candidates(n_docs, args.queries, args.k, args.seed)
    pair_off = np.arange(args.queries + 1, dtype=np.int64) * args.k
    queries = synth.rows(args.seed + 3, scale, 0, args.queries, args.dim)

    scorer = ShardedScorer(shard, doc_bounds, comm)
    scorer.set_queries(queries)
    t_route = time.perf_counter()
    scorer.set_pairs(pair_off, ids, sparse, MODE_NO[args.mode])
    t_route = time.perf_counter() - t_route
    t_route_again = time.perf_counter()  # a batch after the first: the scorer's host and device buffers exist already
    scorer.set_pairs(pair_off, ids, sparse, MODE_NO[args.mode])
    t_route_again = time.perf_counter() - t_route_again
    exact = args.sum == "exact"
    # a whole batch from host buffers to host buffers, every rank (not the metric: the PCIe-inclusive rate next to it):
    # route + upload, local scoring, exchange, owner merge and ordering, download of the owned queries' results
    host_batch = []
    n_own = int(pair_off[query_partition(args.queries, world)[rank + 1]] - pair_off[query_partition(args.queries, world)[rank]])
    keep = (native.pinned_empty(max(1, n_own), np.float32), native.pinned_empty(max(1, n_own), np.float32), native.pinned_empty(max(1, n_own), np.int32))
    for _ in range(3):
        comm.barrier()
        tb = time.perf_counter()
        scorer.set_pairs(pair_off, ids, sparse, MODE_NO[args.mode])
        scorer.run(args.alpha, exact_sum=exact, sort=args.sort)
        scorer.sync()
        scorer.fetch(order=args.sort, out=keep)
        host_batch.append(float(comm.allreduce([time.perf_counter() - tb], native.OP_MAX)[0]))
    t_host_batch = min(host_batch[1:])

    # one step alone, twice: nothing hides its exchange and its merge, and the FIRST one also pays for what the transport sets
    # up lazily (RCCL connects a send/recv channel pair the first time two ranks talk)
    alone = []
    for _ in range(2 + args.warmup):
        comm.barrier()
        ta = time.perf_counter()
        scorer.run(args.alpha, exact_sum=exact, sort=args.sort)
        scorer.sync()
        alone.append(float(comm.allreduce([time.perf_counter() - ta], native.OP_MAX)[0]) * 1e3)
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scorer.run(args.alpha, exact_sum=exact, sort=args.sort)
    t_issued = time.perf_counter() - t0  # the host only queues work: it must stay ahead of the device
    scorer.sync()       # this rank's streams have drained ...
    comm.barrier()      # ... and so have everybody else's
    elapsed = float(comm.allreduce([time.perf_counter() - t0], native.OP_MAX)[0])
    ms_per_step = elapsed * 1e3 / args.steps

    # one more step, alone (no neighbour batch to overlap with): per-rank kernel time from HIP events on the
    # stream the kernel ran on
    scorer.run(args.alpha, exact_sum=exact, sort=args.sort)
    per_rank = gather_stats(comm, scorer.stats())
    row_bytes = args.dim * np.dtype(store).itemsize + 4

    # parity: owned queries' interpolated scores vs the oracle on a small sample (the rank that prints the line) -- of the batch
    # that was replayed above, and once more of the last FRESH batch below
    rehearsal = bool(args.as_rank_of)
    lead = rank == 0 or rehearsal
    base = None

    def check(interp, q_lo, batch_seed):
        nonlocal base
        # the CPU baseline proper is a figure of the N = 1 line; here the oracle only checks a few owned queries
        args.cpu_sample_queries = min(args.cpu_sample_queries, scorer.q_hi - scorer.q_lo, 16 if world > 1 else args.cpu_sample_queries)
        if world > 1:
            args.cpu_seconds = min(args.cpu_seconds, 0.5)
        base, (s_ids, s_sparse, s_interp) = cpu_baseline(args, doc_off, n_docs, scale, q_first=q_lo, batch_seed=batch_seed)
        n_s = s_ids.shape[0]
        got, want = interp[:n_s], s_interp
        held = None
        if rehearsal:  # only the candidates this rank holds were scored: the peers' scores arrive as zeros
            held = (s_ids >= doc_bounds[rank]) & (s_ids < doc_bounds[rank + 1])
            got, want = got[held], want[held]
        ok = np.array_equal(got.view(np.uint32), want.view(np.uint32)) if exact else bool(np.abs(got - want).max() <= 1e-4)
        verdict = ("bit-exact" if exact else "within 1e-4") if ok else "MISMATCH"
        if rehearsal and ok:
            verdict += f" on the {int(held.sum())} of {n_s} sampled pairs this rank holds"
        return verdict

    parity = None
    checking = lead and args.cpu_sample_queries > 0 and scorer.q_hi > scorer.q_lo
    if checking:
        parity = check(scorer.fetch()[1], scorer.q_lo, None)
    # Fresh batches: new code paths over the fabric (a second communicator, exchanges on two lanes).  Whatever happens in there
    # -- an error, a collective that never completes -- the line with the replayed batch's figures still comes out.
    fresh_failed = None
    try:
        fresh = _guarded(lambda: fresh_batches(args, comm, scorer, doc_off, n_docs, doc_bounds, queries, exact, rank, world, query_partition),
                         args.fresh_timeout, "the fresh-batch section")
    except (TimeoutError, native.FFXError, RuntimeError, MemoryError, IndexError, OSError, AssertionError) as e:
        fresh_failed = f"{type(e).__name__}: {e}"
        print(f"rank {rank}: fresh-batch section failed ({fresh_failed}); the line carries the replayed batch only", file=sys.stderr, flush=True)
        fresh = {"line": {"fresh_batch_step_ms": None, "fresh_batch_error": fresh_failed[:300]}, "last_batch_seed": None, "interp": None}
    fresh_parity = None
    if checking and fresh_failed is None:
        interp = fresh.get("interp")  # of the last fresh batch: out of ffx_sharded_score's host buffers, or (rehearsal) out of the device
        if interp is None:
            _, interp, _ = scorer.fetch()
        fresh_parity = check(interp, scorer.q_lo, fresh["last_batch_seed"])
        fresh["line"]["fresh_batch_parity_on_cpu_sample"] = fresh_parity
    failed = parity == "MISMATCH" or fresh_parity == "MISMATCH"
    # every rank learns whether ANY rank's fresh-batch section failed (over the side channel: the job's own communicator may
    # have a call of that section still inside it): then no rank enters the closing collective, and every rank exits 5
    fresh_failed_somewhere = fresh_failed is not None
    votes = getattr(args, "votes", None)
    if votes is not None:
        try:
            fresh_failed_somewhere = bool(votes.allgather_array(np.array([1.0 if fresh_failed is not None else 0.0])).max() > 0)
        except (native.FFXError, TimeoutError, OSError) as e:
            print(f"rank {rank}: the closing vote failed ({type(e).__name__}: {e})", file=sys.stderr, flush=True)
            fresh_failed_somewhere = True
    if lead:
        slowest = max(per_rank, key=lambda t: t["ms_gather_dot"])
        achieved = slowest["n_rows"] * row_bytes / (slowest["ms_gather_dot"] * 1e-3) / 1e9
        dev = native.device_info(local_rank)
        rows_per_rank = [int(t["n_rows"]) for t in per_rank]
        cap = int(dev["total_bytes"] * 0.97 // (args.dim * np.dtype(store).itemsize))
        result = {
            "metric": metric_name(args),
            "value": args.queries / (ms_per_step * 1e-3), "unit": "queries/s",
            "n_gpus": 1 if rehearsal else world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True,
            # a scaling point is only what ran over RCCL on N GPUs: anything else says so here, where a reader of the
            # line (or a driver computing efficiencies from `value`) looks first
            "scaling": ("invalid: single-GPU rehearsal of one rank" if rehearsal else
                        "strong" if (world == 1 or args.transport == "rccl") else "invalid: host transport"),
            "value_is_rccl": bool(args.transport == "rccl" and not rehearsal),
            "vs_baseline": None, "dtype": compute_dtype(args), "data": "synthetic",
            "config": {
                "workload": f"MS MARCO passage-scale synthetic: {args.rows}x{args.dim} {'fp16' if args.store == 'f16' else 'fp32'} index row-sharded by document over "
                            f"{world} {'ranks, of which ONE (rank ' + str(rank) + ') is rehearsed on one GPU' if rehearsal else 'GPUs'} ({n_local} rows on rank {rank}), {args.queries} queries, k={args.k}, Mode.{args.mode}, "
                            f"alpha={args.alpha}, sum={args.sum}; per step: local gather_dot, all-to-all of dense scores to "
                            f"the owning rank ({'ncclSend/ncclRecv full mesh' if args.transport == 'rccl' else 'one copy kernel standing in for the peers' if rehearsal else 'shared-memory transport'}), "
                            f"owner-side interpolation{' + per-query ordering' if args.sort else ''}",
                "rows": args.rows, "dim": args.dim, "queries": args.queries, "k": args.k, "mode": args.mode,
                "layout": args.layout, "sum": args.sum, "sort": bool(args.sort), "pairs": int(pair_off[-1]), "store": args.store,
                "sharding": f"rows by document, {world} ranks; queries owned in {world} contiguous blocks",
                "rows_per_rank": [int(doc_off[doc_bounds[r + 1]] - doc_off[doc_bounds[r]]) for r in range(world)],
                "placement": placement,  # (rank 0's shard: ffx_index_reseat at set-up, see the 1-GPU line)
            },
            "roofline": {
                "bound": "hbm", "kernel": f"gather_dot (variant {int(slowest['kernel_variant'])}), slowest rank of one step",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel_ms_per_rank": [t["ms_gather_dot"] for t in per_rank], "gathered_rows_per_rank": rows_per_rank,
            },
            "transport": args.transport, "rccl_ranks": world if args.transport == "rccl" else 0,
            "comm_env": getattr(args, "comm_knobs", {}),
            "first_step_alone_ms": alone[0], "step_alone_ms": min(alone[1:]),
            "step_alone_note": "one step run by itself, wall clock, max over ranks: local scoring + exchange + owner merge with nothing to hide behind (what the LAST "
                               "batch of a run costs); first_step_alone_ms is the very first one, connection set-up of the transport included",
            **({"rehearsal_exchange": args.rehearsal_exchange} if rehearsal else {}),
            "load_balance": {"gathered_rows_max": max(rows_per_rank), "gathered_rows_mean": float(np.mean(rows_per_rank)),
                             "max_over_mean": max(rows_per_rank) / max(1.0, float(np.mean(rows_per_rank)))},
            "exchange_bytes_per_rank": {"sent": [int(t["send_bytes"]) for t in per_rank], "received": [int(t["recv_bytes"]) for t in per_rank]},
            "local_step_ms_per_rank": [t["ms_total"] for t in per_rank],
            "host_issue_ms_per_step": t_issued * 1e3 / args.steps,
            "exchange_ms_per_rank": [t["ms_exchange"] for t in per_rank], "merge_ms_per_rank": [t["ms_merge"] for t in per_rank],
            "per_rank_note": "HIP events of one step run alone after the timed loop (nothing to overlap with): local = expand + gather_dot on the compute "
                             "stream; exchange and merge (+ per-query order) on the communicator's stream.  In the timed loop batch i's exchange + merge "
                             "overlap batch i+1's scoring, so a step costs about max(local, exchange + merge)",
            "owned_pairs_per_rank": [int(t["own_pairs"]) for t in per_rank],
            "route_and_upload_s": t_route, "route_and_upload_next_batch_s": t_route_again,
            "set_once_host_to_host_batch_ms": t_host_batch * 1e3,
            "set_once_note": "ffx_sharded_set_pairs (every rank uploads and routes the WHOLE batch) + run + fetch, nothing overlapped: the form for a batch that is re-run; "
                             "a fresh batch goes through ffx_sharded_stage_owned / ffx_sharded_score below",
            **fresh["line"],
            "device": dev["name"],
            "index_capacity_rows_per_gpu": cap, "index_capacity_rows_total": world * cap,
        }
        if rehearsal:
            me = per_rank[rank]
            result["rehearsal"] = {
                "as_rank": rank, "of": world,
                "step_ms": ms_per_step, "ideal_ms": me["ms_gather_dot"], "step_over_ideal": ms_per_step / max(me["ms_gather_dot"], 1e-9),
                "local_pairs": int(me["n_pairs"]), "owned_pairs": int(me["own_pairs"]), "owned_queries": int(me["own_queries"]),
                "fresh_batch_step_ms": fresh["line"]["fresh_batch_step_ms"],
                "fresh_batch_step_over_ideal": None if fresh["line"]["fresh_batch_step_ms"] is None else fresh["line"]["fresh_batch_step_ms"] / max(me["ms_gather_dot"], 1e-9),
                "note": "step_ms: K back-to-back steps of this one rank (batch i's exchange + merge overlap batch i+1's scoring); ideal_ms: this "
                        "rank's gather_dot alone (HIP events of a step run by itself).  `value` = queries / step_ms is what N such ranks would "
                        "deliver if the fabric added nothing: a prediction, not a measurement of N GPUs",
            }
        if parity is not None:
            if world == 1:
                result["cpu_baseline"] = base
            result["parity_on_cpu_sample"] = parity
        print(json.dumps(result), flush=True)
    if fresh_failed_somewhere:
        # a call of that section may still be inside the library on its thread: no further collective, no teardown that waits
        # for it.  The line (printed above, with fresh_batch_error) carries the replayed batch only; the status says so: 4 =
        # scores disagree with the oracle, 5 = the fresh-batch section failed or timed out on some rank
        if fresh_failed is None:
            print(f"rank {rank}: another rank's fresh-batch section failed; leaving without the closing collective", file=sys.stderr, flush=True)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(4 if failed else 5)
    failed = bool(comm.allreduce([1.0 if failed else 0.0], native.OP_MAX)[0])
    scorer.close()
    comm.close()
    if votes is not None:
        votes.close()
    if failed:
        sys.exit("sharded GPU scores disagree with the CPU oracle on the sampled queries")


def fresh_batches(args, comm, scorer, doc_off, n_docs, doc_bounds, queries, exact, rank, world, query_partition) -> dict:
    """K DIFFERENT batches back to back, every rank handed only the queries it owns (ffx_sharded_stage_owned): while batch i
    is scored, batch i+1 goes up (this rank's ~P/world candidates), is routed on the device and its ordinals travel to the
    holders.  What the reference does on every call -- a new set of queries (index/base.py:418-429, 445-459) -- and what a
    serving loop would issue.  Then the same through ffx_sharded_score: host buffers in, host buffers out."""
    qb = query_partition(args.queries, world)
    q_lo, q_hi = int(qb[rank]), int(qb[rank + 1])
    n_own_q, n_own = q_hi - q_lo, (q_hi - q_lo) * args.k
    mode = MODE_NO[args.mode]
    rehearsal = bool(args.as_rank_of)
    n_distinct = max(2, min(args.steps, 3))
    own_off = np.arange(n_own_q + 1, dtype=np.int64) * args.k
    own_q = native.pinned_empty((max(n_own_q, 1), args.dim), np.float32)[:n_own_q]
    own_q[...] = queries[q_lo:q_hi]
    replicate = args.fresh_queries == "replicate"
    all_q = None
    if replicate:
        all_q = native.pinned_empty(queries.shape, np.float32)
        all_q[...] = queries
    batches, inboxes, keep_alive = [], [], []
    for b in range(n_distinct):
        seed = args.seed + 1000 * (b + 1)
        ids, sparse = candidates(n_docs, n_own_q, args.k, seed, q_first=q_lo)
        p_ids, p_sp = native.pinned_empty(max(n_own, 1), np.int32)[:n_own], native.pinned_empty(max(n_own, 1), np.float32)[:n_own]
        p_ids[...], p_sp[...] = ids, sparse
        batches.append((seed, p_ids, p_sp))
        if rehearsal and world > 1:
            # what the ranks that do not exist would send this one on the routing lane, in the order the lane receives it:
            # their query vectors, per-query counts, the ordinals of the candidates this rank holds
            parts_cnt, parts_q, parts_ids = [], [], []
            for o in range(world):
                if o == rank:
                    continue
                o_ids, _ = candidates(n_docs, int(qb[o + 1] - qb[o]), args.k, seed, q_first=int(qb[o]))
                mine = (o_ids >= doc_bounds[rank]) & (o_ids < doc_bounds[rank + 1])
                parts_cnt.append(mine.reshape(-1, args.k).sum(axis=1).astype(np.int32))
                parts_q.append(queries[int(qb[o]):int(qb[o + 1])].reshape(-1).view(np.int32))
                parts_ids.append((o_ids[mine] - doc_bounds[rank]).astype(np.int32))
            words = np.concatenate(([] if replicate else parts_q) + parts_cnt + parts_ids)
            box = native.DeviceIndex(max(1, -(-words.shape[0] // 256)), 256, device=comm.device, dtype=np.float32)
            padded = np.zeros(box.capacity * 256, dtype=np.int32)
            padded[:words.shape[0]] = words
            box.upload(padded.view(np.float32).reshape(-1, 256), 0)
            inboxes.append((box.device_ptr(), int(words.shape[0])))
            keep_alive.append(box)

    def stage(b):
        seed, p_ids, p_sp = batches[b % n_distinct]
        if inboxes:
            comm.rehearsal_inbox(*inboxes[b % n_distinct])
        if replicate:
            scorer.stage_queries(all_q)
        scorer.stage_owned(qb, None if replicate else own_q, own_off, p_ids, p_sp, mode)
        return seed

    comm.barrier()
    t_first = time.perf_counter()
    seed = stage(0)  # brings the routing lane up (over RCCL: a second communicator) and allocates the staging buffers
    t_first = float(comm.allreduce([time.perf_counter() - t_first], native.OP_MAX)[0])
    for b in range(args.warmup):
        scorer.run(args.alpha, exact_sum=exact, sort=args.sort)
        seed = stage(b + 1)
    scorer.sync()
    comm.barrier()
    t0 = time.perf_counter()
    t_stage = 0.0
    for b in range(args.steps):
        scorer.run(args.alpha, exact_sum=exact, sort=args.sort)
        if b + 1 < args.steps:
            ts = time.perf_counter()
            seed = stage(args.warmup + b + 1)
            t_stage += time.perf_counter() - ts
    scorer.sync()
    comm.barrier()
    elapsed = float(comm.allreduce([time.perf_counter() - t0], native.OP_MAX)[0])
    st = scorer.stats()
    keys = ("stage_h2d_bytes", "stage_route_send_bytes", "stage_route_recv_bytes", "ms_stage_host", "ms_stage_device", "ms_gather_dot")
    rows = comm.allgather_array(np.array([float(st[k]) for k in keys], dtype=np.float64))
    line = {
        "fresh_batch_step_ms": elapsed * 1e3 / args.steps,
        "fresh_batch_queries_per_s": args.queries / (elapsed / args.steps), "fresh_batch_query_vectors": args.fresh_queries,
        "fresh_first_stage_ms": t_first * 1e3,
        "fresh_batch_note": f"{args.steps} DIFFERENT batches back to back ({n_distinct} distinct candidate sets in page-locked host memory, cycled), every rank handed only "
                            "the queries it owns: upload of the rank's block, holders found on the device, ordinals to the holders in a first all-to-all on the routing "
                            "lane while the previous batch is scored; then local gather_dot, all-to-all of scores, owner-side merge + order.  Results stay in HBM",
        "stage_h2d_bytes_per_rank": [int(r[0]) for r in rows], "stage_route_sent_bytes_per_rank": [int(r[1]) for r in rows],
        "stage_route_received_bytes_per_rank": [int(r[2]) for r in rows],
        "stage_host_ms_per_rank": [r[3] for r in rows], "stage_device_ms_per_rank": [r[4] for r in rows],
        "stage_host_ms_mean_in_loop": t_stage * 1e3 / max(1, args.steps - 1),
        "fresh_batch_gather_dot_ms_per_rank": [r[5] for r in rows],
    }
    if not rehearsal:
        # host buffers -> host buffers (the PCIe-inclusive rate): ffx_sharded_score slices the rank's block and pipelines the slices
        outs = (native.pinned_empty(max(1, n_own), np.float32), native.pinned_empty(max(1, n_own), np.float32), native.pinned_empty(max(1, n_own), np.int32))
        times = []
        for b in range(3):
            sd, p_ids, p_sp = batches[b % n_distinct]
            comm.barrier()
            tb = time.perf_counter()
            scorer.score_owned(qb, own_q, own_off, p_ids, p_sp, mode, args.alpha, exact_sum=exact, sort=args.sort, out=outs)  # (vectors: all-gathered)
            times.append(float(comm.allreduce([time.perf_counter() - tb], native.OP_MAX)[0]))
            seed = sd
        line.update({"host_to_host_batch_ms": min(times[1:]) * 1e3, "host_to_host_queries_per_s": args.queries / min(times[1:]),
                     "host_to_host_note": "ffx_sharded_score: one fresh batch from page-locked host buffers to page-locked host buffers on every rank (each its own block), "
                                          "sliced and pipelined: uploads, routing, scoring, exchange, merge + order and downloads of successive slices overlap (max over ranks)"})
    for box in keep_alive:
        box.close()
    if inboxes:
        comm.rehearsal_inbox(None, 0)
    return {"line": line, "last_batch_seed": seed, "interp": None if rehearsal else outs[1]}


def bench_replicas(args, comm):
    """--sharding queries: the whole index on every GPU, queries split into `world` contiguous blocks,
    no data-path collective.  Same workload and same `value` definition as the row-sharded leg."""
    from fast_forward_amd.sharded import query_partition

    rank, world, local_rank = comm.rank, comm.world, comm.device
    scale = float(1.0 / np.sqrt(args.dim))
    doc_off = doc_layout(args.rows, args.layout, args.seed)
    n_docs = doc_off.shape[0] - 1
    store = np.float16 if args.store == "f16" else np.float32
    index = native.DeviceIndex(args.rows, args.dim, device=local_rank, dtype=store)
    index.fill_synthetic(args.seed, scale, 0, args.rows)
    placement = index.reseat(args.placement_tries)
    index.set_doc_rows(doc_off, np.arange(args.rows, dtype=np.int32))
    index.set_psg_rows(np.arange(args.rows, dtype=np.int32))
    bounds = query_partition(args.queries, world)
    q_lo, q_hi = int(bounds[rank]), int(bounds[rank + 1])
    ids, sparse = candidates(n_docs, q_hi - q_lo, args.k, args.seed, q_first=q_lo)
    pair_off = np.arange(q_hi - q_lo + 1, dtype=np.int64) * args.k
    queries = synth.rows(args.seed + 3, scale, q_lo, q_hi - q_lo, args.dim)
    plan = native.Plan(index)
    plan.set_queries(queries)
    plan.set_pairs(pair_off, ids, sparse, MODE_NO[args.mode])
    exact = args.sum == "exact"
    for _ in range(args.warmup):
        plan.run(args.alpha, exact_sum=exact, sort=args.sort)
    plan.sync()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.run(args.alpha, exact_sum=exact, sort=args.sort)
    plan.sync()
    comm.barrier()
    elapsed = float(comm.allreduce([time.perf_counter() - t0], native.OP_MAX)[0])
    ms_per_step = elapsed * 1e3 / args.steps
    st = plan.stats()
    st.update(own_pairs=st["n_pairs"], own_queries=q_hi - q_lo, send_bytes=0, recv_bytes=0, ms_exchange=0.0, ms_merge=st["ms_reduce"] + st["ms_sort"])
    per_rank = gather_stats(comm, st)
    failed = False
    if rank == 0:
        row_bytes = args.dim * np.dtype(store).itemsize + 4
        slowest = max(per_rank, key=lambda t: t["ms_gather_dot"])
        achieved = slowest["n_rows"] * row_bytes / (slowest["ms_gather_dot"] * 1e-3) / 1e9
        result = {
            "metric": metric_name(args),
            "value": args.queries / (ms_per_step * 1e-3), "unit": "queries/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": compute_dtype(args), "data": "synthetic",
            "config": {
                "workload": f"MS MARCO passage-scale synthetic: {args.rows}x{args.dim} index replicated on {world} GPUs, "
                            f"{args.queries} queries split into {world} blocks, k={args.k}, Mode.{args.mode}, alpha={args.alpha}, "
                            f"sum={args.sum}; no exchange (comparison point of the row-sharded layout)",
                "rows": args.rows, "dim": args.dim, "queries": args.queries, "k": args.k, "mode": args.mode,
                "layout": args.layout, "sum": args.sum, "sort": bool(args.sort), "pairs": args.queries * args.k, "store": args.store,
                "sharding": f"queries: {world} replicas, contiguous query blocks", "placement": placement,
            },
            "roofline": {
                "bound": "hbm", "kernel": f"gather_dot (variant {int(slowest['kernel_variant'])}), slowest rank of the last step",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "kernel_ms_per_rank": [t["ms_gather_dot"] for t in per_rank], "gathered_rows_per_rank": [int(t["n_rows"]) for t in per_rank],
            },
            "transport": args.transport,
        }
        if args.cpu_sample_queries > 0:
            args.cpu_sample_queries = min(args.cpu_sample_queries, q_hi - q_lo)
            base, (s_ids, s_sparse, s_interp) = cpu_baseline(args, doc_off, n_docs, scale)
            _, interp, _ = plan.fetch(interp=True)
            n_s = s_ids.shape[0]
            ok = np.array_equal(interp[:n_s].view(np.uint32), s_interp.view(np.uint32)) if exact else bool(np.abs(interp[:n_s] - s_interp).max() <= 1e-4)
            result["cpu_baseline"] = base
            result["parity_on_cpu_sample"] = ("bit-exact" if exact else "within 1e-4") if ok else "MISMATCH"
            failed = not ok
        print(json.dumps(result), flush=True)
    failed = bool(comm.allreduce([1.0 if failed else 0.0], native.OP_MAX)[0])
    plan.close()
    comm.close()
    if failed:
        sys.exit("GPU scores disagree with the CPU oracle on the sampled queries")


def timed(fn, reps: int) -> float:
    """Best of `reps` wall-clock times of fn(), in ms."""
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def side_regions(args, index, plan, queries, pair_off, ids, sparse, exact: bool) -> dict:
    """the regions around the headline, measured in this same run.  (i) was timed by
    the caller; here (ii) host buffers in -> host buffers out through ffx_score, and what the box
    streams (the same kernel reading every row of the index exactly once)."""
    mode = MODE_NO[args.mode]
    out = {}
    # (i) without the ordering
    plan.run(args.alpha, exact_sum=exact, sort=False)
    plan.sync()
    out["i_device_ms"] = timed(lambda: (plan.run(args.alpha, exact_sum=exact, sort=False), plan.sync()), 5)
    # (ii) ffx_score, scores + order: caller's arrays pageable, then page-locked
    n = int(pair_off[-1])
    # the caller's own (ordinary) result arrays, written before: a C caller re-uses its buffers; fresh ones would add
    # a page fault per 4 KB of the 420 MB of results to the measurement
    mine = (np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.int32))
    out["ii_pageable_ms"] = timed(lambda: native.score(index, queries, pair_off, ids, mode, sparse, args.alpha, exact_sum=exact, sort=args.sort, out=mine), 3)
    del mine
    p_q = native.pinned_empty(queries.shape, np.float32)
    p_ids, p_sp = native.pinned_empty(n, np.int32), native.pinned_empty(n, np.float32)
    p_q[...], p_ids[...], p_sp[...] = queries, ids, sparse
    outs = (native.pinned_empty(n, np.float32), native.pinned_empty(n, np.float32), native.pinned_empty(n, np.int32))
    run_ii = lambda: native.score(index, p_q, pair_off, p_ids, mode, p_sp, args.alpha, exact_sum=exact, sort=args.sort, out=outs)  # noqa: E731
    run_ii()
    out["ii_pinned_ms"] = timed(run_ii, 4)
    out["ii_pinned_queries_per_s"] = args.queries / (out["ii_pinned_ms"] * 1e-3)
    out["ii_note"] = ("ffx_score: candidate CSR + sparse scores + query vectors in host memory -> dense + interpolated scores"
                      f"{' + per-query order' if args.sort else ''} in host memory ({(n * (8 + 4 * (3 if args.sort else 2)) + queries.nbytes) / 1e6:.0f} MB over PCIe "
                      "per step); slices of whole queries rotate through three plans, their kernels run in order on one stream, so uploads, kernels and downloads of successive slices overlap")
    # what the box streams: every row once, in order and in a random permutation
    if args.store != "pq" and args.layout == "single":
        row_bytes = args.dim * index.dtype.itemsize + 4
        n_rows = args.rows
        q4 = np.ascontiguousarray(queries[:4])
        off4 = np.array([0, n_rows // 4, n_rows // 2, 3 * (n_rows // 4), n_rows], dtype=np.int64)
        sweep = native.Plan(index)
        sweep.set_queries(q4)
        ceil = {}
        for label, rows in (("in_order", np.arange(n_rows, dtype=np.int32)),
                            ("permuted", np.random.default_rng(0).permutation(n_rows).astype(np.int32))):
            sweep.set_pairs(off4, rows, None, 1)
            ms = []
            for _ in range(4):
                sweep.run(0.0, exact_sum=exact)
                ms.append(sweep.stats()["ms_gather_dot"])
            ceil[label] = n_rows * row_bytes / float(np.mean(ms[1:])) / 1e6
        sweep.close()
        out["stream_ceiling_GBs"] = ceil
        out["stream_ceiling_note"] = "the gather_dot kernel itself over every index row exactly once (no reuse possible): the practical ceiling of this box for this access pattern"
    return out


def api_context(args, index, doc_off, scale: float, exact: bool) -> "dict | None":
    """What regions (iii) share: the bench's index adopted by an ``InMemoryIndex`` under string document ids, the query
    encoder (a table of vectors), the candidate lists of the workload."""
    from fast_forward_amd.encoder import ArrayEncoder
    from fast_forward_amd.index import InMemoryIndex, Mode

    if not hasattr(InMemoryIndex, "adopt"):
        return None
    nq = min(args.api_queries, args.queries)
    n_docs = doc_off.shape[0] - 1
    t0 = time.perf_counter()
    # names as Python strings, one object per document / query -- what a frame read from a run file holds (pandas'
    # readers keep one object per distinct value of a column), and what a frame built from arrays of names holds
    doc_names = np.array([f"D{i}" for i in range(n_docs)], dtype=object)
    api = InMemoryIndex.adopt(index, doc_ids=doc_names, doc_off=doc_off, mode=Mode[args.mode], exact_sum=exact)
    adopt_s = time.perf_counter() - t0
    ids, sparse = candidates(n_docs, nq, args.k, args.seed)
    qvecs = synth.rows(args.seed + 3, scale, 0, nq, args.dim)
    q_names = np.array([f"q{i:06d}" for i in range(nq)], dtype=object)
    api.query_encoder = ArrayEncoder(dict(zip(q_names.tolist(), qvecs)))
    return {"api": api, "nq": nq, "n_docs": n_docs, "doc_names": doc_names, "q_names": q_names, "ids": ids, "sparse": sparse, "adopt_s": adopt_s}


def api_region(args, ctx: dict) -> dict:
    """Region (iii): the public API on the same index -- a first-stage ``Ranking`` of STRING ids in,
    ``InMemoryIndex.rerank`` (= ``ranking.interpolate(index(ranking), alpha)`` with the epilogue on
    the device) out, id resolution and frame construction included."""
    import pandas as pd

    from fast_forward_amd import Ranking

    api, nq, n_docs, doc_names, q_names, ids, sparse = (ctx[k] for k in ("api", "nq", "n_docs", "doc_names", "q_names", "ids", "sparse"))
    out = {"iii_adopt_ids_s": ctx["adopt_s"], "iii_ids": n_docs}
    t0 = time.perf_counter()
    frame = pd.DataFrame({"q_id": np.repeat(q_names, args.k), "id": doc_names[ids], "score": sparse,
                          "query": np.repeat(q_names, args.k)})
    out["iii_build_frame_s"] = time.perf_counter() - t0  # (the bench's own input, not the product)
    t0 = time.perf_counter()
    first = Ranking(frame, name="first-stage", copy=False)
    out["iii_ranking_constructor_s"] = time.perf_counter() - t0
    out["iii_build_first_stage_ranking_s"] = out["iii_build_frame_s"] + out["iii_ranking_constructor_s"]
    # COLD: the first call on a fresh Ranking -- distinct ids looked up in the index, candidate ordinals derived, plans and
    # result buffers allocated, upload, device, download, result gathers (nothing of it cached yet)
    t0 = time.perf_counter()
    final = api.rerank(first, args.alpha)
    out["iii_rerank_first_call_s"] = time.perf_counter() - t0
    out["iii_first_call_breakdown"] = dict(getattr(api, "last_timings", {}))
    out["iii_queries_per_s_cold"] = nq / (out["iii_ranking_constructor_s"] + out["iii_rerank_first_call_s"])
    # SECOND call on the same ranking: the resolved candidate list is kept with the ranking, but the first result is still
    # alive, so this result needs two MORE page-locked blocks of the result pool (~60 ms to make, once per number of results
    # alive at a time)
    t0 = time.perf_counter()
    final = api.rerank(first, args.alpha)
    out["iii_rerank_second_call_s"] = time.perf_counter() - t0
    out["iii_second_call_breakdown"] = dict(getattr(api, "last_timings", {}))
    # WARM: the same ranking again (another alpha on the same run), steady state: the best of three further calls
    warm = []
    for _ in range(3):
        t0 = time.perf_counter()
        final = api.rerank(first, args.alpha)
        warm.append((time.perf_counter() - t0, dict(getattr(api, "last_timings", {}))))
    out["iii_rerank_s"], out["iii_warm_call_breakdown"] = min(warm, key=lambda w: w[0])
    out["iii_rerank_again_s"] = out["iii_rerank_s"]  # (the name round 4's first lines used)
    # what the breakdown does not see: binding the result (the previous one is released: its blocks go back to the result pool)
    out["iii_rerank_outside_the_breakdown_s"] = out["iii_rerank_s"] - out["iii_warm_call_breakdown"].get("total_s", 0.0)
    out["iii_pairs"] = int(nq * args.k)
    out["iii_queries_per_s_rerank"] = nq / out["iii_rerank_s"]
    t0 = time.perf_counter()
    n_final = len(final._df)  # results come back as integer codes; the pandas frame of strings is built on first use
    out["iii_result_frame_on_first_use_s"] = time.perf_counter() - t0
    out["iii_result_frame_s_per_million_rows"] = out["iii_result_frame_on_first_use_s"] / max(1e-9, nq * args.k / 1e6)
    # the worst frame for the constructor: every cell its own string object (nothing to share), on the first queries only
    nq3 = min(nq, 400)
    own = {c: np.array([(x + " ")[:-1] for x in frame[c].to_numpy()[:nq3 * args.k].tolist()], dtype=object) for c in ("q_id", "id", "query")}
    t0 = time.perf_counter()
    Ranking(pd.DataFrame(dict(own, score=sparse[:nq3 * args.k])), name="first-stage", copy=False)
    out["iii_ranking_constructor_one_object_per_cell_s"], out["iii_ranking_constructor_one_object_per_cell_rows"] = time.perf_counter() - t0, int(nq3 * args.k)
    del own
    # the reference's own two-call idiom on the same run: Index.__call__, then Ranking.interpolate (join-free for a
    # ranking that came out of index(first); combined and re-ordered by host threads)
    api(first)
    t0 = time.perf_counter()
    dense = api(first)
    t1 = time.perf_counter()
    final2 = first.interpolate(dense, args.alpha)
    out["iii_call_s"], out["iii_interpolate_s"] = t1 - t0, time.perf_counter() - t1
    # (which route the interpolate took: "queries_numbered_s" = from the dense scores index(first) left in HBM, no second gather)
    out["iii_interpolate_breakdown"] = dict(getattr(api, "last_timings", None) or {})
    again = []
    for _ in range(3):  # the blocks of the result pool exist after the first calls (how many results are alive at once decides when)
        t1 = time.perf_counter()
        final2 = first.interpolate(dense, args.alpha)
        again.append(time.perf_counter() - t1)
    out["iii_interpolate_again_s"] = min(again)
    out["iii_queries_per_s_call_then_interpolate"] = nq / (out["iii_call_s"] + out["iii_interpolate_s"])
    same = (final._coded is not None and final2._coded is not None and np.array_equal(final._coded["id_codes"], final2._coded["id_codes"])
            and np.array_equal(final._coded["score"].view(np.uint32), final2._coded["score"].view(np.uint32)))
    out["iii_rerank_equals_call_then_interpolate"] = bool(same) and n_final == nq * args.k
    # the general host join (rankings that are not each other's image), on the first queries only: ~0.2 us per pair
    nq2 = min(nq, 400)
    sub = Ranking(frame.iloc[:nq2 * args.k], name="first-stage", copy=False)
    other = Ranking(api(sub)._df.copy(), name="dense", copy=False)  # a fresh frame: nothing links it to `sub`
    t0 = time.perf_counter()
    joined = sub.interpolate(other, args.alpha)
    out["iii_general_join_interpolate_s"], out["iii_general_join_pairs"] = time.perf_counter() - t0, int(nq2 * args.k)
    out["iii_general_join_equals_rerank"] = bool(joined == api.rerank(sub, args.alpha))
    out["iii_note"] = (f"{nq} queries x k={args.k} from a pandas frame of string ids: the Ranking constructor (iii_ranking_constructor_s), then "
                       f"Index.rerank COLD (iii_rerank_first_call_s: first call on that ranking, distinct-id look-up and allocations included) and "
                       f"WARM (iii_rerank_s: the same ranking again, best of three calls after the second, which is iii_rerank_second_call_s); Index.__call__ followed by "
                       f"Ranking.interpolate (iii_interpolate_s: first time, new result blocks; iii_interpolate_again_s: steady state); the general host join on {nq2} queries")
    ctx["frame"] = frame  # (region iii_fresh starts from the same frame)
    return out


def fresh_region(args, ctx: dict) -> dict:
    """Region (iii) for a run the index has NEVER seen -- what every real call is (the reference scores a new ranking on each
    call, index/base.py:418-429; README.md:34-52): nothing of the ranking is cached, every route below builds its own
    ``Ranking`` and pays for the id look-up.  6980 x k rows each:
      a_*        columns in -> ``Ranking`` -> ``Index.rerank`` -> ``Ranking.columns()`` (arrays out; no frame of strings)
                 from a pandas frame of Python strings, Arrow string arrays, fixed-width bytes, pre-coded integers
      b_file     ``Ranking.from_file`` -> ``attach_queries`` -> ``index()`` -> ``interpolate`` -> ``save`` (ranking.py:388-409, :348-366)
      c_rerank_file  ``Index.rerank_file``: the same, file to file, no Ranking in between
    Every stage is timed; the stages of a route sum to its total.  All routes must give the same rows, scores and order."""
    import shutil
    import tempfile

    import pandas as pd

    from fast_forward_amd import Ranking
    from fast_forward_amd import ranking as ranking_module

    api, nq, n_docs, doc_names, q_names, ids = (ctx[k] for k in ("api", "nq", "n_docs", "doc_names", "q_names", "ids"))
    k, alpha = args.k, args.alpha
    # a first-stage run lists every query's candidates best first
    sparse = np.ascontiguousarray(-np.sort(-ctx["sparse"].reshape(nq, k), axis=1).reshape(-1))
    queries = dict(zip(q_names.tolist(), q_names.tolist()))
    out = {"rows": int(nq * k), "queries": int(nq), "distinct_ids": int(np.unique(ids).shape[0])}

    def stages(route: dict) -> dict:
        route["total_s"] = float(sum(v for key, v in route.items() if key.endswith("_s") and isinstance(v, float)))
        return route

    reference = {}  # the first route's result (copies: the result itself goes back to the pool like a served one would)
    same = {"arrays": True}

    def rerank_and_hand_out(first, route: dict):
        t0 = time.perf_counter()
        final = api.rerank(first, alpha)
        route["rerank_first_call_s"] = time.perf_counter() - t0
        route["rerank_breakdown"] = dict(getattr(api, "last_timings", {}))
        t0 = time.perf_counter()
        cols = final.columns()
        route["result_columns_s"] = time.perf_counter() - t0
        compare(cols)

    def compare(cols: dict) -> None:
        """Every route must hand out the rows, scores and order of the first one (untimed)."""
        if not reference:
            reference.update(score=np.array(cols["score"]), id_codes=np.array(cols["id_codes"]), packed=cols["id_names"].packed, q_names=list(cols["q_names"]),
                             names=cols["id_names"])
            return
        ok = np.array_equal(cols["score"].view(np.uint32), reference["score"].view(np.uint32)) and cols["q_names"] == reference["q_names"]
        if cols["id_names"].packed == reference["packed"]:  # (one numbering rule: equal tables, so the codes must agree)
            ok = ok and np.array_equal(cols["id_codes"], reference["id_codes"])
        else:
            rows = np.random.default_rng(0).integers(0, len(cols["id_codes"]), size=4096)
            ok = ok and all(cols["id_names"][int(cols["id_codes"][r])] == reference["names"][int(reference["id_codes"][r])] for r in rows)
        same["arrays"] = same["arrays"] and bool(ok)

    # ---- (a) from columns -------------------------------------------------------------------------------
    frame = ctx.pop("frame", None)
    if frame is None or not np.array_equal(frame["score"].to_numpy(), sparse):
        frame = pd.DataFrame({"q_id": np.repeat(q_names, k), "id": doc_names[ids], "score": sparse, "query": np.repeat(q_names, k)})
    route = {}
    t0 = time.perf_counter()
    first = Ranking(frame, name="first-stage", copy=False)
    route["ranking_constructor_s"] = time.perf_counter() - t0
    route["constructor_breakdown"] = dict(ranking_module.BUILD_TIMINGS)
    rerank_and_hand_out(first, route)
    out["a_frame_of_python_strings"] = stages(route)
    del first
    forms = {}
    try:
        import pyarrow as pa

        forms["a_arrow_strings"] = (pa.array(frame["q_id"].to_numpy(), type=pa.string()), pa.array(frame["id"].to_numpy(), type=pa.string()))
    except ImportError:
        pass
    forms["a_fixed_width_bytes"] = (np.repeat(q_names.astype("S"), k), doc_names.astype("S")[ids])
    forms["a_pre_coded_integers"] = ((np.repeat(np.arange(nq, dtype=np.int32), k), q_names), (ids, doc_names.astype("S")))
    del frame
    for label, (q_form, id_form) in forms.items():
        route = {}
        t0 = time.perf_counter()
        first = Ranking.from_columns(q_form, id_form, sparse, name="first-stage", queries=queries)
        route["from_columns_s"] = time.perf_counter() - t0
        route["constructor_breakdown"] = dict(ranking_module.BUILD_TIMINGS)
        rerank_and_hand_out(first, route)
        out[label] = stages(route)
        del first
    forms.clear()

    # ---- (b), (c) file to file --------------------------------------------------------------------------
    work = Path(tempfile.mkdtemp(prefix="ffx_fresh_"))
    try:
        run_in = work / "first_stage.tsv"
        pair_off = np.arange(nq + 1, dtype=np.int64) * k
        assert native.save_run_coded(run_in, q_names.tolist(), pair_off, ids, doc_names, sparse, "bm25")
        out["run_file_MB"] = run_in.stat().st_size / 1e6
        route = {}
        t = [time.perf_counter()]
        first = Ranking.from_file(run_in)
        t.append(time.perf_counter())
        route["from_file_stages"] = native.stage_report()
        first = first.attach_queries(queries)
        t.append(time.perf_counter())
        dense = api(first)
        t.append(time.perf_counter())
        final_b = first.interpolate(dense, alpha)
        t.append(time.perf_counter())
        final_b.save(work / "out_b.tsv")
        t.append(time.perf_counter())
        route["save_stages"] = native.stage_report()
        route.update(dict(zip(("from_file_s", "attach_queries_s", "call_s", "interpolate_s", "save_s"), np.diff(t).tolist())))
        out["b_file"] = stages(route)
        out["b_file"]["GB_per_s"] = {"from_file": out["run_file_MB"] / 1e3 / route["from_file_s"], "save": (work / "out_b.tsv").stat().st_size / 1e9 / route["save_s"]}
        route = {}
        t0 = time.perf_counter()
        api.rerank_file(run_in, work / "out_c.tsv", alpha, queries)
        route["rerank_file_s"] = time.perf_counter() - t0
        route["stages"] = dict(getattr(api, "last_file_timings", {}))
        out["c_rerank_file"] = stages(route)
        # ---- all routes give the same run -----------------------------------------------------------------
        import filecmp

        same_files = filecmp.cmp(work / "out_b.tsv", work / "out_c.tsv", shallow=False)
        compare(final_b.columns())
        out["outputs_identical"] = {"file_route_equals_rerank_file_bytes": bool(same_files), "every_column_route_equals_the_file_route": same["arrays"]}
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out["queries_per_s"] = {key: nq / val["total_s"] for key, val in out.items() if isinstance(val, dict) and "total_s" in val}
    return out


def measure_traffic(args) -> dict:
    """roofline.traffic of the 1-GPU line, measured in this run: HBM bytes per launch of the dominant kernel
    from the PMC counters, as MI355X_MICROARCH.md prescribes -- each counter in its own rocprofv3 --pmc pass
    (never combined with tracing), FETCH_SIZE is in KiB and counts 64-B units where gfx950 moves 128-B
    ones (x2; calibrated on launches with known traffic: tools/pmc_calibrate.py), WRITE_SIZE in KiB as it is.
    Each pass is a CHILD process (`rocprofv3 ... -- python3 bench.py --steps 2 ...` on the same workload),
    started before this process has touched the GPU.  Anything going wrong leaves traffic null and says why."""
    import csv
    import shutil
    import tempfile

    if args.traffic_bytes is not None:
        return {"traffic": args.traffic_bytes, "traffic_source": "--traffic-bytes (rocprofv3 --pmc pass of this command)"}
    if not args.traffic:
        return {"traffic": None, "traffic_source": None}
    tool = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if tool is None:
        return {"traffic": None, "traffic_source": "rocprofv3 not found"}
    child = [sys.executable, str(Path(__file__).resolve()), "--steps", "2", "--warmup", "1", "--no-regions", "--cpu-sample-queries", "0", "--no-traffic",
             "--rows", str(args.rows), "--dim", str(args.dim), "--queries", str(args.queries), "--k", str(args.k), "--mode", args.mode,
             "--layout", args.layout, "--sum", args.sum, "--store", args.store, "--alpha", str(args.alpha), "--seed", str(args.seed),
             "--sort" if args.sort else "--no-sort"]
    want = "pq_adc" if (args.store == "pq" and args.sum == "exact") else "gather_dot"
    got, launches = {}, 0
    t0 = time.perf_counter()
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(prefix="ffx_pmc_", dir="/tmp") as out:
                env = dict(os.environ, TMPDIR="/tmp")
                # its own process group, so that a pass that hangs can be ended together with the program under it
                proc = subprocess.Popen([tool, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--"] + child, cwd="/tmp", env=env,
                                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
                try:
                    _, err = proc.communicate(timeout=150)
                except subprocess.TimeoutExpired:
                    import signal

                    os.killpg(proc.pid, signal.SIGKILL)  # the group started two lines up
                    proc.communicate()
                    return {"traffic": None, "traffic_source": f"rocprofv3 --pmc {counter} did not finish within 150 s"}
                if proc.returncode != 0:
                    return {"traffic": None, "traffic_source": f"rocprofv3 --pmc {counter} exited with {proc.returncode}: {err.strip()[-200:]}"}
                per_dispatch = {}
                for f in Path(out).rglob("*counter_collection.csv"):
                    for row in csv.DictReader(open(f)):
                        if want in row["Kernel_Name"] and row["Counter_Name"] == counter:
                            key = (f.name, row["Dispatch_Id"])
                            per_dispatch[key] = per_dispatch.get(key, 0.0) + float(row["Counter_Value"])
                if not per_dispatch:
                    return {"traffic": None, "traffic_source": f"no {want} dispatch in the {counter} pass"}
                got[counter] = sum(per_dispatch.values()) / len(per_dispatch)
                launches = len(per_dispatch)
    except (OSError, subprocess.SubprocessError, KeyError, ValueError) as e:
        return {"traffic": None, "traffic_source": f"PMC pass failed: {type(e).__name__}: {str(e)[:200]}"}
    fetch, write = 2.0 * 1024.0 * got["FETCH_SIZE"], 1024.0 * got["WRITE_SIZE"]
    return {"traffic": fetch + write,
            "traffic_source": f"measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate child passes of this workload, "
                              f"{launches} launches of the {want} kernel each, {time.perf_counter() - t0:.0f} s); FETCH_SIZE KiB x 1024 x 2 (gfx950) + WRITE_SIZE KiB x 1024",
            "traffic_fetch_bytes": fetch, "traffic_write_bytes": write}


def bench_single(args):
    pmc = measure_traffic(args)  # child processes, before this one touches the GPU
    native.require_device()
    dev = native.device_info(0)
    scale = float(1.0 / np.sqrt(args.dim))  # scores O(1): 1e-4 absolute ~ 1e-4 relative
    doc_off = doc_layout(args.rows, args.layout, args.seed)
    n_docs = doc_off.shape[0] - 1

    t_setup = time.perf_counter()
    store = {"f16": np.float16, "pq": np.uint8}.get(args.store, np.float32)
    pq_m, pq_ks = 8, 256
    index = native.DeviceIndex(args.rows, pq_m if args.store == "pq" else args.dim, device=0, dtype=store)
    if args.store == "pq":  # synthetic codebook [M][Ks][Ds] and uniform random codes
        index.set_codebook(synth.rows(args.seed + 9, scale, 0, pq_m * pq_ks, args.dim // pq_m).reshape(pq_m, pq_ks, -1))
    index.fill_synthetic(args.seed, scale, 0, args.rows)
    placement = index.reseat(args.placement_tries)  # (index set-up, like the fill: not in the timed region)
    index.set_doc_rows(doc_off, np.arange(args.rows, dtype=np.int32))
    index.set_psg_rows(np.arange(args.rows, dtype=np.int32))
    ids, sparse = candidates(n_docs, args.queries, args.k, args.seed)
    pair_off = np.arange(args.queries + 1, dtype=np.int64) * args.k
    queries = synth.rows(args.seed + 3, scale, 0, args.queries, args.dim)
    plan = native.Plan(index)
    plan.set_queries(queries)
    plan.set_pairs(pair_off, ids, sparse, MODE_NO[args.mode])
    t_setup = time.perf_counter() - t_setup

    exact = args.sum == "exact"
    for _ in range(args.warmup):
        plan.run(args.alpha, exact_sum=exact, sort=args.sort)
    plan.sync()
    # the K steps are queued back to back (no wait in between, as a serving loop would issue them); every step's
    # gather_dot is bracketed by its own pair of HIP events on the plan's stream, read after the timed region
    t0 = time.perf_counter()
    for _ in range(args.steps):
        plan.run(args.alpha, exact_sum=exact, sort=args.sort)
    plan.sync()
    elapsed = time.perf_counter() - t0
    ms_per_step = elapsed * 1e3 / args.steps
    kern_ms = plan.recent_gather_ms(min(args.steps, 256)).astype(np.float64).tolist()
    st = plan.stats()  # the other stages: HIP events of the last step
    stage_ms = {key: st[key] for key in ("ms_expand", "ms_reduce", "ms_sort", "ms_total")}

    # per gathered row, payload + int32 row id
    row_bytes = (pq_m if args.store == "pq" else args.dim * np.dtype(store).itemsize) + 4
    kern_avg_ms = float(np.mean(kern_ms))
    achieved = st["n_rows"] * row_bytes / (kern_avg_ms * 1e-3) / 1e9
    result = {
        "metric": metric_name(args),
        "value": args.queries / (ms_per_step * 1e-3),
        "unit": "queries/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": compute_dtype(args), "data": "synthetic",
        "value_region": "(i): inputs and outputs resident in HBM when the timed region starts, per-query ordering included.  The round brief's measurement "
                        "section fixes this (\"`value` is whole-job throughput with inputs already resident in HBM when the timed region starts (if the boundary "
                        "hands over host buffers, note the PCIe-inclusive rate in DESIGN.md -- it is never `value`)\": quoted in DESIGN.md section 7); SURVEY 8(d) "
                        "names region (ii), host buffers -> host buffers through ffx_score, as its headline: that rate is `value_host_to_host` right below, "
                        "and (iii), through Index.rerank / Index.__call__ + Ranking.interpolate, is in `regions`",
        "config": {
            "workload": f"MS MARCO passage-scale synthetic: {args.rows}x{args.dim} { {'f16': 'fp16', 'pq': 'PQ(M=8,Ks=256)'}.get(args.store, 'fp32') } InMemoryIndex in HBM, "
                        f"{args.queries} queries, k={args.k}, Mode.{args.mode}, alpha={args.alpha}, "
                        f"{'1 passage/doc' if args.layout == 'single' else '1..8 passages/doc'}, sum={args.sum}"
                        f"{', per-query ordering on the device' if args.sort else ''}; inputs and outputs resident in HBM",
            "rows": args.rows, "dim": args.dim, "queries": args.queries, "k": args.k, "mode": args.mode,
            "layout": args.layout, "sum": args.sum, "sort": bool(args.sort), "pairs": st["n_pairs"], "gathered_rows": st["n_rows"],
            "sharding": "none (1 GPU)", "index_bytes": args.rows * (pq_m if args.store == "pq" else args.dim * np.dtype(store).itemsize), "store": args.store,
            # where the rows lie: the index was moved into the fastest of `candidates` allocations when it was set up (ffx_index_reseat =
            # InMemoryIndex.consolidate(); probe_ms: a 2 M-row random gather over each; chosen 0 = the one hipMalloc gave first)
            "placement": placement,
        },
        "roofline": {
            "bound": "hbm", "kernel": f"gather_dot (variant {st['kernel_variant']})",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            # product-quantized rows: what limits the kernel is the REQUEST rate of a random gather, not bytes -- every 8-byte row
            # costs one 128-byte read request (rocprofv3 --pmc TCC_EA0_RDREQ_sum: 1.03 per row, none of them 32-byte ones,
            # the same request the fp32 kernel issues 24 of per 3072-byte row: profiles/r04/pmc_pq_requests.json)
            **({"bound": "hbm, as 128-byte read requests: a random gather of 8-byte rows moves one line per row (10.7 x the algorithmic bytes)",
                "achieved": st["n_rows"] * 128 / (kern_avg_ms * 1e-3) / 1e9, "frac": st["n_rows"] * 128 / (kern_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "achieved_is": "gathered rows x 128 B (one read request each) / kernel time, table build of the exact ADC included",
                "algorithmic_achieved": achieved, "algorithmic_frac": achieved / HBM_PEAK_GBS} if args.store == "pq" else {}),
            # HBM bytes per launch from the PMC counters: measured by child rocprofv3 passes of this workload (measure_traffic)
            **pmc,
            **({"traffic_over_algorithmic": pmc["traffic"] / (st["n_rows"] * row_bytes)} if pmc.get("traffic") else {}),
            "algorithmic_bytes_per_launch": st["n_rows"] * row_bytes, "kernel_ms_avg": kern_avg_ms,
            "kernel_ms_min": float(np.min(kern_ms)), "kernel_ms_max": float(np.max(kern_ms)),
        },
        "stages_ms": {"expand": stage_ms["ms_expand"], "gather_dot": kern_avg_ms, "reduce_interp": stage_ms["ms_reduce"],
                      "sort": stage_ms["ms_sort"], "device_total": stage_ms["ms_total"]},
        "device": dev["name"], "setup_s": t_setup,
        "index_capacity_rows_per_gpu": int(dev["total_bytes"] * 0.97 // (pq_m if args.store == "pq" else args.dim * np.dtype(store).itemsize)),
    }

    ok = True
    if args.cpu_sample_queries > 0:
        base, (s_ids, s_sparse, s_interp) = cpu_baseline(args, doc_off, n_docs, scale)
        result["cpu_baseline"] = base
        # the same sample through the GPU path must agree with the oracle (bit-exact in exact mode)
        _, interp, _ = plan.fetch(interp=True)
        n_s = s_ids.shape[0]
        assert np.array_equal(ids[:n_s], s_ids)
        if exact:
            ok = np.array_equal(interp[:n_s].view(np.uint32), s_interp.view(np.uint32))
        else:
            ok = bool(np.abs(interp[:n_s] - s_interp).max() <= 1e-4)
        result["parity_on_cpu_sample"] = "bit-exact" if (ok and exact) else ("within 1e-4" if ok else "MISMATCH")
    if ok and args.regions:
        regions = {"i_device_sorted_ms" if args.sort else "i_device_ms": ms_per_step}
        regions.update(side_regions(args, index, plan, queries, pair_off, ids, sparse, exact))
        if args.api_queries > 0 and args.store == "f32" and args.layout == "single":
            ctx = api_context(args, index, doc_off, scale, exact)
            if ctx is not None:
                regions.update(api_region(args, ctx))
                regions["iii_fresh"] = fresh_region(args, ctx)
                ctx["api"].release()
        result["regions"] = regions
        # SURVEY 8(d)'s headline region, next to `value`: CSR in host memory -> scores + order in host memory (page-locked buffers)
        result["value_host_to_host"] = args.queries / (regions["ii_pinned_ms"] * 1e-3)
        result["value_host_to_host_over_value"] = result["value_host_to_host"] / result["value"]
        # the public API against region (ii), warm: Index.rerank; Index.__call__ followed by Ranking.interpolate (the reference's
        # two-call form) with the result pool warm (`again`) and as first measured (`first`: new result blocks are page-locked)
        if "iii_rerank_again_s" in regions and int(regions.get("iii_pairs", 0)) == int(pair_off[-1]):
            ii_s = regions["ii_pinned_ms"] * 1e-3
            result["api_over_host_to_host"] = {
                "rerank_warm": regions["iii_rerank_again_s"] / ii_s,
                "call_then_interpolate_warm": (regions["iii_call_s"] + regions["iii_interpolate_again_s"]) / ii_s,
                "call_then_interpolate_first": (regions["iii_call_s"] + regions["iii_interpolate_s"]) / ii_s,
                "rerank_first_call_s": regions["iii_rerank_first_call_s"]}
    result["value_hbm_resident"] = result["value"]
    print(json.dumps(result), flush=True)
    if not ok:
        sys.exit("GPU scores disagree with the CPU oracle on the sampled queries")


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))  # no GPU call has been made in this process
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world
    if world > 1 or args.force_sharded or args.as_rank_of:
        bench_sharded(args, rank, world)
        if " unavailable" in str(args.transport):
            # a communicator that never came up may still have a thread inside RCCL; do not wait for it at exit
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        return None
    return bench_single(args)


if __name__ == "__main__":
    main()
