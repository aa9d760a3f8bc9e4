/*
 * ffx.h -- C ABI of libffx: the MI355X (gfx950) Fast-Forward re-ranking path.
 *
 * The reference (mrjleo/fast-forward-indexes v0.8.0) is pure Python and has no FFI; its
 * seam for this path is the abstract `Index` class (src/fast_forward/index/base.py:30):
 * `Index.__call__` (:389-469) reaches storage through `_get_vectors` (:258-277, :291) and
 * does the arithmetic in `_compute_scores` (:279-314); `Ranking.interpolate`
 * (src/fast_forward/ranking.py:293-326) adds the sparse scores.  Every entry point below
 * names the reference lines it replaces.  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types; no exceptions cross the ABI
 *   - every function returns an FFX_* status (0 = ok); ffx_last_error() gives the message
 *     of the calling thread's last failure
 *   - the caller owns every host buffer for the duration of the call; the library owns
 *     device memory behind the opaque handles and retains no host pointer after return
 *   - a handle is not thread-safe; distinct handles are independent
 *   - ids are integers here: a *document ordinal* names the rows of one document (in
 *     insertion order), a *passage ordinal* names one row.  String ids stay in the host
 *     language (reference: the dicts at src/fast_forward/index/memory.py:46-47).
 *
 * This header is the STABLE contract: what INTEGRATION.md binds and tests/c_abi/ uses from plain C.  The helpers the
 * Python layer of this repository uses on top of it (threaded host gathers / joins / orderings over NumPy arrays, string
 * numbering, the building blocks ffx_sharded and the streamed index are made of, test-data generators) live in
 * ffx_private.h: exported by the same library, no stability promise.
 */
#ifndef FFX_H_
#define FFX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FFX_ABI_VERSION 3  /* 3: ffx_shard_stats grew (the stage_* fields of the owner-routed form); 2: (ms_exchange, ms_merge); everything else is additive */

/* status codes */
#define FFX_OK 0
#define FFX_BAD_ARG (-1)
#define FFX_OOM (-2)
#define FFX_HIP_ERROR (-3)
#define FFX_RCCL_ERROR (-4) /* a collective of the multi-GPU path failed (RCCL, or a peer rank that went away) */
#define FFX_NOT_FOUND (-5)  /* an ordinal names no row: the reference's IndexError, index/util.py:38-39 */
#define FFX_NO_DEVICE (-6)

/* ranking modes: values of fast_forward.index.Mode, index/base.py:18-24 */
#define FFX_MODE_PASSAGE 1
#define FFX_MODE_MAXP 2
#define FFX_MODE_FIRSTP 3
#define FFX_MODE_AVEP 4

/* storage type of index rows.  The reference keeps whatever dtype it is given
 * (index/memory.py:80-83); float32 queries x float16 rows promote to float32 products in NumPy,
 * so half rows are widened exactly and scored with the same fp32 arithmetic */
#define FFX_DTYPE_F32 0
#define FFX_DTYPE_F16 1
/* product-quantizer codes, one uint8 per subspace (the reference's quantized index: add() stores
 * quantizer.encode(vectors), index/base.py:253, and scoring decodes first, index/base.py:292-293).
 * Rows have `dim` = M codes; the codebook [M][Ks][Ds] fp32 comes through ffx_index_set_codebook and
 * query vectors have M*Ds elements. */
#define FFX_DTYPE_PQ8 2

/* ffx_plan_run flags */
#define FFX_FLAG_EXACT_SUM 1u /* NumPy summation order: scores bit-identical to index/base.py:303 */
#define FFX_FLAG_SORT 2u      /* also produce the per-query order of ranking.py:115-117 */
#define FFX_FLAG_SORT_FF 4u   /* with SORT: order by the dense score even if sparse scores are set */
/* ffx_plan_early_stopping only: evaluate the continue-criterion's bound alpha*s_last + (1-alpha)*max_dense in float32,
 * as NumPy >= 2 (NEP 50) evaluates the reference's expression (index/base.py:353-355: Python float x np.float32 scalar).
 * Default (flag absent): in float64 and compared with the float32 cut-off-th score as a double -- what the reference
 * computes under its pinned numpy>=1.24,<2 (pyproject.toml:16).  The two differ when a decision sits within float32
 * rounding of the bound (always, e.g., at cut-off 1 after a first depth of 1). */
#define FFX_FLAG_ES_BOUND_F32 8u
/* ffx_score_ranked only: the dense scores of the call (in the caller's candidate order) and the carried words stay in HBM with
 * the index (8 bytes per candidate) until the next such call, ffx_index_drop_kept or ffx_index_destroy: ffx_rank_kept. */
#define FFX_FLAG_KEEP_DENSE 16u

/* ffx_plan_device_ptr selectors */
#define FFX_BUF_FF 0       /* float[P]   aggregated dense scores                         */
#define FFX_BUF_INTERP 1   /* float[P]   interpolated scores (valid if sparse was given) */
#define FFX_BUF_ORDER 2    /* int32[P]   per-query order (valid after FFX_FLAG_SORT)     */
#define FFX_BUF_QUERIES 3  /* float[Q*d] query vectors                                   */
#define FFX_BUF_PAIR_IDS 4 /* int32[P]   candidate ordinals                              */
#define FFX_BUF_SPARSE 5   /* float[P]   first-stage scores                              */

typedef struct ffx_index ffx_index;
typedef struct ffx_plan ffx_plan;

typedef struct ffx_stats {
  int64_t n_queries;
  int64_t n_pairs;           /* P: (query, candidate) pairs                                   */
  int64_t n_rows;            /* R: gathered index rows (= P unless MAXP/AVEP with >1 passage) */
  double algorithmic_bytes;  /* R*(d*e+4) + P*4*(1+2*interp) + Q*d*4 + (Q+1)*8, e = bytes/element */
  float ms_expand;           /* pairs -> row list (count, scan, expand)                       */
  float ms_gather_dot;       /* the gather + inner-product kernel                             */
  float ms_reduce;           /* segment max/first/mean + interpolation                        */
  float ms_sort;             /* per-query ordering (0 unless FFX_FLAG_SORT)                   */
  float ms_total;            /* first launch to last completion of ffx_plan_run               */
  int32_t kernel_variant;    /* which gather_dot instantiation ran      */
  int32_t reserved;
} ffx_stats;

/* ---- library / device ------------------------------------------------------------- */
int ffx_abi_version(void);
const char* ffx_last_error(void);
int ffx_device_count(int* count);
int ffx_device_info(int device, char* name, size_t name_cap, int64_t* total_bytes,
                    int64_t* free_bytes, int32_t* compute_units);

/* ---- HBM-resident index: replaces InMemoryIndex storage, index/memory.py:23-131 -------
 * One fp32 matrix of `capacity_rows` x `dim` in device memory, rows in insertion order
 * (row r of the reference's chunk list, memory.py:100-119, is row r here). */
int ffx_index_create(int device, int64_t capacity_rows, int32_t dim, ffx_index** out); /* fp32 rows */
int ffx_index_create_typed(int device, int64_t capacity_rows, int32_t dim, int32_t dtype, ffx_index** out);
int ffx_index_dtype(const ffx_index* idx, int32_t* dtype);
/* codewords of a product quantizer (quantizer/nanopq.py:49-56): float[m][ks][ds], m = row length */
int ffx_index_set_codebook(ffx_index* idx, const float* codewords, int32_t m, int32_t ks, int32_t ds);
/* dimensionality of query vectors: dim for float rows, m*ds for product-quantized rows */
int ffx_index_query_dim(const ffx_index* idx, int32_t* dim);
int ffx_index_destroy(ffx_index* idx);
int ffx_index_dim(const ffx_index* idx, int32_t* dim);
int ffx_index_capacity(const ffx_index* idx, int64_t* rows);
/* device address of row 0 (rows are contiguous, in the index dtype): zero-copy consumers / producers on the same GPU */
int ffx_index_device_ptr(const ffx_index* idx, void** ptr);
/* Where the rows lie in HBM (closest in the reference: InMemoryIndex.consolidate, index/memory.py:121-131 -- "I am done adding,
 * lay the vectors out for look-ups").  How fast a random gather over the index runs belongs to the ALLOCATION: copies of one
 * 27 GB index side by side in one process score the same batch 3 - 6 % apart, each within 0.3 % of its own figure.
 * ffx_index_reseat allocates up to `tries` (0 .. 7; -1: as many as half of the free memory holds, at most 7) further blocks
 * of the index's size (fewer when memory runs out), times a
 * short random gather of whole rows over each of them (a copy of the rows in each: memory nobody has written reads faster)
 * and over the block in use, keeps the fastest, frees the others and waits until the device is at rest again (see settle_s).
 * Transient memory: up to `tries` x the index; time: tens of ms per block + the wait.
 * An index below 256 MiB is left where it is.  No call on the index may be in flight; addresses from ffx_index_device_ptr
 * are stale afterwards.  `report` (may be NULL) says what was measured. */
typedef struct ffx_reseat_report {
  int32_t candidates;  /* blocks timed, the one in use included (1: nothing was tried) */
  int32_t chosen;      /* 0: the block the index had; k: the k-th new one */
  float probe_ms[8];   /* time of the probe over each block */
  int64_t bytes;       /* size of a block */
  float settle_s;      /* for four seconds after blocks of this size are freed, gathers run 1 - 3 % slower: the call waits them out */
  float probe_ms_settled; /* ... and until the probe over the chosen block reads this (within 1 % of its figure as a candidate; 8 s at most) */
} ffx_reseat_report;
int ffx_index_reseat(ffx_index* idx, int32_t tries, ffx_reseat_report* report);
/* copy host rows into rows [row_offset, row_offset+n_rows): InMemoryIndex._add, memory.py:100-119 */
int ffx_index_upload(ffx_index* idx, const void* host_rows, int64_t row_offset, int64_t n_rows); /* rows in the index dtype */
/* rows scattered over the equally sized chunks of a (memory-mapped) file -> device rows
 * [dst_row, dst_row+n_rows): rows[i] names stored row rows[i] % chunk_rows of chunk
 * rows[i] / chunk_rows (the chunk maps of OnDiskIndex, index/disk.py:100-121).  Host threads
 * (n_threads; 0 = half the cores) fill page-locked staging while the previous block is copied. */
int ffx_index_upload_rows(ffx_index* idx, const void* const* chunks, int64_t n_chunks, int64_t chunk_rows,
                          const int64_t* rows, int64_t n_rows, int64_t dst_row, int32_t n_threads);
/* device copy between two indexes on the same device (growing an index) */
int ffx_index_copy_rows(ffx_index* dst, const ffx_index* src, int64_t n_rows);
/* read rows back: InMemoryIndex._get_vectors, memory.py:139-140 / index/util.py:97-98 */
int ffx_index_gather(ffx_index* idx, const int32_t* rows, int64_t n_rows, void* host_out); /* in the index dtype */
/* id structure: the two dicts of memory.py:46-47 as arrays.
 *   doc ordinal o -> rows doc_rows[doc_off[o] .. doc_off[o+1])   (insertion order)
 *   psg ordinal o -> row psg_row[o]                               (-1: no such passage) */
int ffx_index_set_doc_rows(ffx_index* idx, const int64_t* doc_off, int64_t n_docs,
                           const int32_t* doc_rows, int64_t n_doc_rows);
int ffx_index_set_psg_rows(ffx_index* idx, const int32_t* psg_row, int64_t n_psgs);

/* ---- scoring plan: replaces Index._compute_scores, index/base.py:279-314, and the
 * arithmetic of Ranking.interpolate, ranking.py:319 ------------------------------------
 * A plan owns the device workspace of one batch of queries.  Typical use:
 *   set_queries, set_pairs  (host -> HBM; validates ordinals; FFX_NOT_FOUND + *bad_pair)
 *   run                     (device only, asynchronous: expand -> gather_dot -> reduce [-> sort])
 *   sync, fetch             (HBM -> host)
 * A plan can be re-run with another alpha / flags without re-uploading. */
int ffx_plan_create(ffx_index* idx, ffx_plan** out);
int ffx_plan_destroy(ffx_plan* plan);
/* query vectors [n_queries, dim] fp32, row q = query number q (q_no of index/base.py:418-421) */
int ffx_plan_set_queries(ffx_plan* plan, const float* host_queries, int64_t n_queries);
/* the same for query vectors that are already in DEVICE memory of the plan's GPU (a query encoder running
 * on the GPU hands over its output without a trip through the host): device-to-device copy on the plan's stream */
int ffx_plan_set_queries_device(ffx_plan* plan, const float* dev_queries, int64_t n_queries);
/* candidate lists in CSR form: pairs pair_off[q] .. pair_off[q+1] belong to query q and name
 * ordinal pair_ids[p] (document ordinal, or passage ordinal for FFX_MODE_PASSAGE); sparse[p] is
 * the first-stage score (may be NULL).  get_indices, index/util.py:12-42, runs here. */
int ffx_plan_set_pairs(ffx_plan* plan, const int64_t* pair_off, int64_t n_queries,
                       const int32_t* pair_ids, const float* sparse, int mode, int64_t* bad_pair);
int ffx_plan_run(ffx_plan* plan, double alpha, uint32_t flags);
/* ---- early stopping: replaces Index._early_stopping, index/base.py:316-387 ------------------
 * Candidates of a query stand in first-stage rank order.  set_pairs_lazy uploads the lists without
 * validating ordinals (use -1 for an id the host could not resolve): as in the reference, an unknown
 * id is an error only if it is reached (FFX_NOT_FOUND + *bad_pair).  early_stopping scores depth
 * intervals [a, b) over the sorted depths >= cutoff (a repeated depth ends the loop, as in the reference); after
 * the first interval a query continues only while its cutoff-th best interpolated score is below
 * alpha * (first-stage score of its last scored candidate) + (1 - alpha) * (its best dense score so far),
 * evaluated in float64 (NumPy < 2 scalar promotion, the reference's pinned range) or, with
 * FFX_FLAG_ES_BOUND_F32, in float32 (NumPy >= 2).  All per-query state stays on the device;
 * per interval two counters come back to size the launches.  Afterwards ffx_plan_fetch gives the
 * dense and interpolated scores of ALL pairs, NaN where a candidate was never scored. */
int ffx_plan_set_pairs_lazy(ffx_plan* plan, const int64_t* pair_off, int64_t n_queries, const int32_t* pair_ids,
                            const float* sparse, int mode);
int ffx_plan_early_stopping(ffx_plan* plan, int32_t cutoff, double alpha, const int32_t* depths, int32_t n_depths,
                            uint32_t flags, int64_t* bad_pair, int64_t* n_scored_pairs, int32_t* n_intervals);
/* after ffx_plan_early_stopping: the scored candidates only, compacted on the device -- positions in the candidate
 * list (ascending) and dense scores; *n_scored = how many there are, up to `cap` of them are written */
int ffx_plan_fetch_scored(ffx_plan* plan, int64_t cap, int32_t* out_pos, float* out_ff, int64_t* n_scored);
int ffx_plan_sync(ffx_plan* plan);
/* any of the outputs may be NULL */
int ffx_plan_fetch(ffx_plan* plan, float* out_ff, float* out_interp, int32_t* out_order);
int ffx_plan_stats(ffx_plan* plan, ffx_stats* out);
/* raw device pointers (score exchange between ranks, zero-copy consumers); valid until the
 * next set_* call on the plan */
int ffx_plan_device_ptr(ffx_plan* plan, int which, void** ptr, int64_t* n_elems);
/* the plan's HIP stream (hipStream_t) so callers can order their own work after run */
int ffx_plan_stream(ffx_plan* plan, void** stream);

/* run the plan on a caller-owned HIP stream (hipStream_t; e.g. the stream a collective library
 * orders its work on) instead of the plan's own; NULL restores the plan's own stream */
int ffx_plan_set_stream(ffx_plan* plan, void* stream);

/* ---- multi-GPU: one process per GPU, row-sharded index ------------------------------------------
 * The reference is single-process; this is the sharded form of the same path
 * (Index._compute_scores, index/base.py:279-314, per shard; Ranking.interpolate + sort,
 * ranking.py:293-326, at the query's owner).  Shard = a contiguous range of documents, so
 * MaxP/FirstP/AveP stay rank-local; queries are replicated; the ONE exchange per batch is an
 * all-to-all-v of fp32 dense scores (every rank sends each owner its slice directly).
 *
 * ffx_comm: the communicator.  Bootstrap: one rank calls ffx_comm_unique_id and hands the
 * FFX_COMM_ID_BYTES to the others by any means (a file, a socket, MPI, a torch store); then every
 * rank calls ffx_comm_init (collective). */
#define FFX_COMM_RCCL 0 /* ncclSend/ncclRecv full mesh over xGMI on a HIP stream; one GPU per rank        */
#define FFX_COMM_HOST 1 /* staged through POSIX shared memory: ranks may share a GPU, or have none (-1) */
/* ONE process standing in for rank `rank` of `world` ranks that do not exist: a timing rehearsal of one rank's step
 * on one GPU (bench.py --as-rank-of N).  Control collectives return this rank's own contribution in every slot; the
 * exchange keeps the slice the rank sends itself and moves, through this GPU's memory, the bytes the peers' slices
 * would move (one kernel: sent slices are copied into a sink, received slices arrive as zeros out of a scratch
 * source).  Scores of candidates other ranks would hold are therefore 0 -- never use this transport for results. */
#define FFX_COMM_REHEARSAL 2
#define FFX_COMM_ID_BYTES 128
#define FFX_OP_SUM 0
#define FFX_OP_MAX 1
#define FFX_OP_MIN 2
typedef struct ffx_comm ffx_comm;
int ffx_comm_unique_id(int transport, void* id /* FFX_COMM_ID_BYTES out */);
int ffx_comm_init(int transport, int device, int rank, int world, const void* id, ffx_comm** out);
int ffx_comm_destroy(ffx_comm* comm);
int ffx_comm_rank(const ffx_comm* comm, int* rank, int* world);
int ffx_comm_stream(ffx_comm* comm, void** stream); /* hipStream_t the exchange is ordered on */
/* control plane (host values, small): blocking collectives */
int ffx_comm_barrier(ffx_comm* comm);
int ffx_comm_allreduce_f64(ffx_comm* comm, double* inout, int32_t n, int op);
int ffx_comm_allgather(ffx_comm* comm, const void* mine, int64_t bytes, void* all /* world*bytes, rank order */);
/* all-to-all-v of fp32 scores between DEVICE buffers: slice j of send_dev (send_counts[j] floats,
 * slices in rank order) goes to rank j; recv_dev receives recv_counts[i] floats from rank i, in
 * rank order.  RCCL: asynchronous on `stream` (NULL: the communicator's).  Host transport: blocks
 * until the scores have arrived. */
int ffx_comm_exchange(ffx_comm* comm, const float* send_dev, const int64_t* send_counts, float* recv_dev,
                      const int64_t* recv_counts, void* stream);
/* the same between HOST buffers (shared-memory transport: needs no GPU; RCCL: staged through device
 * scratch -- for scores a streamed index produced on the host) */
int ffx_comm_exchange_host(ffx_comm* comm, const float* send, const int64_t* send_counts, float* recv,
                           const int64_t* recv_counts);

/* routing, host only: from the CSR of ALL queries (pair_ids = GLOBAL document ordinals) what rank
 * `rank` scores and what it finalises.  Shards are the document ranges doc_bounds[r]..doc_bounds[r+1];
 * or, with holder != NULL, pair p is held by rank holder[p] and pair_ids[p] is already the
 * holder-local ordinal (doc_bounds may be NULL).  Queries are owned in `world` contiguous blocks
 * (ffx_query_partition).  Outputs (caller-allocated):
 *   local_pair_off[Q+1], local_pair_ids[<= P] (may be NULL: sizes only)  this rank's CSR over all queries
 *   send_counts[world]   local scores going to each owner (contiguous slices of the local list)
 *   recv_counts[world]   scores of the owned queries arriving from each rank
 *   recv_slot[own pairs] (may be NULL) position of owned candidate p in the received buffer
 *   own[4]               first owned query, one past the last, first owned pair, number of owned pairs
 * FFX_NOT_FOUND (+ *bad_pair) if a pair names a document outside every shard. */
int ffx_query_partition(int64_t n_queries, int world, int64_t* bounds /* world+1 */);
int ffx_route(const int64_t* pair_off, int64_t n_queries, const int32_t* pair_ids, const int32_t* holder,
              const int64_t* doc_bounds, int rank, int world, int64_t* local_pair_off, int32_t* local_pair_ids,
              int64_t* send_counts, int64_t* recv_counts, int32_t* recv_slot, int64_t* own, int64_t* bad_pair);

/* ffx_sharded: one rank of the row-sharded index = local plans + exchange + owner merge.  All ranks
 * make the same calls with the same arguments (SPMD).  Scoring runs on a compute stream, the exchange of scores on the
 * communicator's stream, and the owner-side merge of batch i is queued on the compute stream BEHIND the scoring of batch
 * i+1 -- by the NEXT ffx_sharded_run, or by ffx_sharded_sync / _fetch for the last batch -- so back-to-back runs hide
 * the exchange and a step costs about the local scoring.  Results are therefore valid only after ffx_sharded_sync or
 * ffx_sharded_fetch (ffx_sharded_fetch_async hands out the batch BEFORE the one run last); synchronising
 * ffx_comm_stream is not enough, and ffx_sharded_device_ptr names the buffers of the batch merged last.
 *
 * Two ways to hand a batch over:
 *   ffx_sharded_set_pairs[_held]   every rank passes the CSR of ALL queries (set once, run many times: another alpha, a
 *                                  timing loop); each rank uploads and routes all P candidates
 *   ffx_sharded_stage_owned        every rank passes only the block of queries it OWNS (~P/world candidates): the owners
 *                                  find the holders on the device and the candidate ordinals travel to them in a first
 *                                  all-to-all (4 bytes per pair, same mesh as the scores) on a routing lane of the
 *                                  communicator, WHILE the previous batch is being scored.  What the reference does on
 *                                  every call -- score a new set of queries (index/base.py:418-429, 445-459) -- at 1/world
 *                                  of the intake cost per rank.  ffx_sharded_score is the host-to-host call built on it. */
typedef struct ffx_sharded ffx_sharded;
typedef struct ffx_shard_stats {
  ffx_stats local;      /* the local ffx_plan_run of the last step (kernel times, gathered rows) */
  int64_t own_queries;  /* queries this rank finalises   */
  int64_t own_pairs;    /* and their candidates          */
  int64_t send_bytes;   /* scores sent to / received from OTHER ranks per step */
  int64_t recv_bytes;
  int32_t rank, world, transport, reserved;
  float ms_exchange;    /* last step, HIP events on the communicator's stream: the all-to-all of scores */
  float ms_merge;       /* ... and the owner-side scatter + interpolation + per-query order            */
  /* the last ffx_sharded_stage_owned (0 for a batch set with ffx_sharded_set_pairs) */
  int64_t stage_h2d_bytes;        /* host -> device: this rank's block only (offsets, ordinals, first-stage scores, query vectors) */
  int64_t stage_route_send_bytes; /* routing lane, to / from OTHER ranks: per-query counts, ordinals, query vectors               */
  int64_t stage_route_recv_bytes;
  float ms_stage_host;            /* wall-clock of the call (it waits for its own routing stream three times)                     */
  float ms_stage_device;          /* first upload to last routing kernel, HIP events on the routing stream                        */
} ffx_shard_stats;
/* shard: this rank's rows (document ordinals relative to the shard); doc_bounds[world+1]: first
 * global document ordinal of every shard (NULL: pairs come with explicit holders, below) */
int ffx_sharded_create(ffx_index* shard, ffx_comm* comm, const int64_t* doc_bounds, ffx_sharded** out);
int ffx_sharded_destroy(ffx_sharded* s);
int ffx_sharded_set_queries(ffx_sharded* s, const float* host_queries, int64_t n_queries);
/* the CSR of ALL queries with GLOBAL document ordinals, as for ffx_plan_set_pairs */
int ffx_sharded_set_pairs(ffx_sharded* s, const int64_t* pair_off, int64_t n_queries, const int32_t* pair_ids,
                          const float* sparse, int mode, int64_t* bad_pair);
/* pair p is held by rank holder[p]; local_ids[p] = its ordinal on that rank (read where holder[p] == rank) */
int ffx_sharded_set_pairs_held(ffx_sharded* s, const int64_t* pair_off, int64_t n_queries, const int32_t* holder,
                               const int32_t* local_ids, const float* sparse, int mode, int64_t* bad_pair);
/* A FRESH batch, routed by the ranks that own its queries.  q_bounds[world + 1]: rank r owns queries q_bounds[r] ..
 * q_bounds[r+1] of the batch (the same array on every rank; any block sizes).  The rest describes THIS rank's block only:
 * own_queries [Qo, dim] (NULL: the replicated vectors of ffx_sharded_set_queries stay), own_pair_off[Qo + 1] starting at
 * 0, own_pair_ids = global document ordinals (shards = the doc_bounds ranges) or, with own_holder != NULL, the ordinal
 * on rank own_holder[p]; own_sparse may be NULL.  Collective, and may be called while the previous batch runs (three
 * batches rotate: staged, running, merging); ffx_sharded_run then runs the batch staged last.  Every rank returns the
 * same status: FFX_NOT_FOUND when any rank's candidate names an ordinal without rows (*bad_pair is set on the rank that
 * owns it: a position in ITS list).  Per-rank host->device bytes are proportional to the block, not to the batch.
 * Every own_* array has been read when the call returns: the caller may refill it for the next batch at once. */
int ffx_sharded_stage_owned(ffx_sharded* s, const int64_t* q_bounds, const float* own_queries, const int64_t* own_pair_off,
                            const int32_t* own_pair_ids, const int32_t* own_holder, const float* own_sparse, int mode, int64_t* bad_pair);
/* Shards whose rows only their holder can name (rows placed by a hash of the document id: every rank interns the ids of its own
 * rows and no others): the candidates of ffx_sharded_stage_owned / _score then travel as CODES -- any numbering of the distinct
 * ids that every rank agrees on, e.g. a ranking's id codes -- with explicit holders, and the holder translates them after they
 * have arrived: ordinal_of_code[c] = this rank's ordinal of code c (-1: not held here).  The table stays in device memory
 * until it is replaced; NULL: candidates are ordinals again.  (The reference has one id table per process:
 * index/memory.py:46-47; here every rank has its own.) */
int ffx_sharded_set_code_table(ffx_sharded* s, const int32_t* ordinal_of_code, int64_t n_codes);
/* The other way to hand over the query vectors of the batch about to be staged: ALL n_queries of them, on every rank (each
 * rank uploads Q * dim * 4 bytes over its own link; nothing is exchanged), then ffx_sharded_stage_owned with own_queries =
 * NULL.  Asynchronous (the vectors must stay put until the stage call returns).  Which of the two is cheaper depends on the
 * node: the all-gather keeps the intake per rank proportional to its block but runs as a copy kernel beside gather_dot,
 * the replicated upload costs PCIe time only (MEASUREMENTS.md section 4 has both measured). */
int ffx_sharded_stage_queries(ffx_sharded* s, const float* all_queries, int64_t n_queries);
int ffx_sharded_run(ffx_sharded* s, double alpha, uint32_t flags); /* asynchronous */
/* the results of the batch merged last -- after ffx_sharded_run that is the batch BEFORE the one it ran -- copied out on a
 * stream of their own (truly asynchronous into page-locked buffers); complete after ffx_sharded_sync */
int ffx_sharded_fetch_async(ffx_sharded* s, float* out_ff, float* out_interp, int32_t* out_order);
/* One batch from host buffers to host buffers (arguments as ffx_sharded_stage_owned; results of the OWNED queries,
 * first-stage order, out_order = positions in the rank's own list): the block is cut into n_slices slices of whole queries
 * (0: by size; the same on every rank) that go through stage / run / fetch_async back to back, so uploads, routing,
 * scoring, exchange, merge and downloads of successive slices overlap.  The multi-GPU form of ffx_score. */
int ffx_sharded_score(ffx_sharded* s, const int64_t* q_bounds, const float* own_queries, const int64_t* own_pair_off,
                      const int32_t* own_pair_ids, const int32_t* own_holder, const float* own_sparse, int mode, double alpha,
                      uint32_t flags, int32_t n_slices, float* out_ff, float* out_interp, int32_t* out_order, int64_t* bad_pair);
int ffx_sharded_sync(ffx_sharded* s);
/* results of the OWNED queries, first-stage order (ffx_sharded_owned tells which) */
int ffx_sharded_fetch(ffx_sharded* s, float* out_ff, float* out_interp, int32_t* out_order);
int ffx_sharded_owned(const ffx_sharded* s, int64_t* q_lo, int64_t* q_hi, int64_t* first_pair, int64_t* n_pairs);
int ffx_sharded_stats(ffx_sharded* s, ffx_shard_stats* out);
int ffx_sharded_device_ptr(ffx_sharded* s, int which, void** ptr, int64_t* n_elems);

/* ---- stand-alone epilogue: the reduce / interpolate / sort stages of ffx_plan_run applied to row
 * scores held by the caller (host arrays in, host arrays out).  A streamed (on-disk) index scores
 * its rows window by window and aggregates once at the end: candidate p owns row scores
 * seg_off[p] .. seg_off[p+1] (index/base.py:306-312, ranking.py:319, ranking.py:115-117);
 * seg_off == NULL means one row per candidate. */
int ffx_aggregate(int device, const float* row_scores, const int64_t* seg_off, int64_t n_pairs, int mode,
                  const float* sparse, double alpha, const int64_t* pair_off, int64_t n_queries, uint32_t flags,
                  float* out_ff, float* out_interp, int32_t* out_order);

/* ---- host natives: id tables and TREC run files ---------------------------------------------
 * ffx_ids: insertion-ordered string -> ordinal table; replaces the per-id dict look-ups of
 * get_indices (index/util.py:29-41).  Strings cross the ABI as arrays of fixed-width, NUL-padded
 * byte fields (numpy dtype 'S<width>'). */
typedef struct ffx_ids ffx_ids;
int ffx_ids_create(ffx_ids** out);
int ffx_ids_destroy(ffx_ids* table);
int ffx_ids_count(const ffx_ids* table, int64_t* n);
int ffx_ids_intern(ffx_ids* table, const char* strs, int32_t width, int64_t n, int32_t* out_ordinals);
/* FFX_NOT_FOUND (+ position of the first unknown string) if any is absent; absent -> -1 */
int ffx_ids_lookup(const ffx_ids* table, const char* strs, int32_t width, int64_t n, int32_t* out_ordinals,
                   int64_t* first_missing);
int ffx_ids_get(const ffx_ids* table, int32_t ordinal, const char** str, int32_t* len);

/* ffx_run: a TREC run file (q_id Q0 id rank score name) parsed straight into the CSR of
 * ffx_plan_set_pairs, and written back; replaces Ranking.from_file / cut / save
 * (ranking.py:388-409, :279-291, :348-366) on the file-to-file path.  Candidate ids are resolved
 * against `ids` (which must outlive the run); an unknown id is FFX_NOT_FOUND (+ its line). */
typedef struct ffx_run ffx_run;
/* `ids` NULL: no id table to resolve against (Ranking.from_file): the candidate ids are kept as
 * strings (ffx_run_ids_blob) and the q_id / id columns are classified the way pandas.read_csv
 * would type them (ffx_run_flags), so a caller can tell whether the plain strings are what the
 * reference's reader would have produced. */
#define FFX_TOKEN_INT 1    /* an integer literal that prints back unchanged                      */
#define FFX_TOKEN_NUMBER 2 /* another number or boolean: re-printed differently, or only column-wide */
#define FFX_TOKEN_NA 4     /* one of pandas' NA words: the row would be dropped                  */
#define FFX_TOKEN_TEXT 8   /* anything else: the column stays text                               */
#define FFX_RUN_SHORT_LINE 1 /* a line with 5 fields (no run name)                               */
#define FFX_RUN_LONG_LINE 2  /* a line with more than 6 fields                                   */
#define FFX_RUN_QUOTES 4     /* a double quote somewhere in the file                             */
int ffx_run_open(const char* path, const ffx_ids* ids, ffx_run** out, int64_t* bad_line);
int ffx_run_close(ffx_run* run);
int ffx_run_counts(const ffx_run* run, int64_t* n_queries, int64_t* n_pairs);
int ffx_run_query_id(const ffx_run* run, int64_t q, const char** str, int32_t* len);
int ffx_run_tag(const ffx_run* run, const char** str, int32_t* len);
/* cutoff > 0: keep each query's `cutoff` best-scoring candidates first (in place) */
int ffx_run_export(ffx_run* run, int64_t cutoff, int64_t* pair_off, int32_t* pair_ids, float* sparse, int64_t* n_pairs);
int ffx_run_write(const ffx_run* run, const char* path, const float* scores, const int32_t* order, int64_t keep,
                  const char* tag);

/* one-shot, host buffers in, host buffers out.  A small batch of single-row candidates is submitted in one go
 * and waited for once.  A large batch is cut into slices of whole queries that rotate through three plans kept
 * with the index: the slices' kernels run in order on one stream while the next slice is uploaded and the
 * previous one's results leave (asynchronously into page-locked buffers, through helper threads into ordinary
 * ones), so uploads, scoring and downloads overlap for either kind of memory.  Results do not depend on the
 * slicing; stats are summed over the slices.  Environment: FFX_DEBUG_VARIANT=score_slice_pairs=N overrides the slice size
 * (pairs), FFX_SCORE_TRACE=1 prints the host-side timeline of the slices on stderr. */
int ffx_score(ffx_index* idx, const float* host_queries, int64_t n_queries,
              const int64_t* pair_off, const int32_t* pair_ids, const float* sparse, int mode,
              double alpha, uint32_t flags, float* out_ff, float* out_interp, int32_t* out_order,
              int64_t* bad_pair, ffx_stats* stats);

/* ffx_score with the ordering APPLIED on the device: out_scores[i] / out_carry[i] = the ranking score (interpolated when
 * sparse is given, else dense) and the carried word of the candidate at position i of the per-query order -- the run as
 * Ranking.__init__ lays it out (ranking.py:115-117), with whatever the caller numbers its candidates with (the id codes of
 * a run) permuted along, so the host gathers nothing afterwards.  carry / out_carry may be NULL; out_order (positions in
 * the caller's list, as ffx_score) may be NULL.  FFX_FLAG_SORT is implied. */
int ffx_score_ranked(ffx_index* idx, const float* host_queries, int64_t n_queries, const int64_t* pair_off,
                     const int32_t* pair_ids, const float* sparse, const int32_t* carry, int mode, double alpha, uint32_t flags,
                     float* out_scores, int32_t* out_carry, int32_t* out_order, int64_t* bad_pair, ffx_stats* stats);

/* Ranking.interpolate on the result of Index.__call__ (ranking.py:293-326 after index/base.py:389-469) without scoring twice:
 * a call of ffx_score_ranked with FFX_FLAG_KEEP_DENSE leaves its dense scores in HBM; ffx_index_kept says under which token
 * (0: nothing is kept) and for how many queries / candidates; ffx_rank_kept then produces, for the SAME candidate lists
 * (pair_off must be the one of the keeping call: checked) and any first-stage scores and alpha, what ffx_score_ranked with those
 * arguments would -- interpolated scores and carried words in final order, out_order optional -- by the interpolation and the
 * ordering alone: no row of the index is read.  sparse may be NULL (the dense scores, ordered).  A token dies with the next
 * keeping call on the index; rows uploaded in between do not change what is kept (the caller decides whether it still wants it). */
int ffx_index_kept(ffx_index* idx, uint64_t* token, int64_t* n_queries, int64_t* n_pairs);
int ffx_index_drop_kept(ffx_index* idx);
int ffx_rank_kept(ffx_index* idx, uint64_t token, int64_t n_queries, const int64_t* pair_off, const float* sparse, double alpha,
                  uint32_t flags, float* out_scores, int32_t* out_carry, int32_t* out_order, ffx_stats* stats);

/* ---- page-locked host memory ------------------------------------------------------------
 * Host buffers handed to the calls above may live anywhere; when they are page-locked the
 * copies to and from the device run at full link rate and overlap the kernels (ffx_score
 * pipelines slices of the batch).  ffx_host_alloc returns such memory, ffx_host_pin locks a
 * range the caller already owns (e.g. a NumPy array) until ffx_host_unpin. */
int ffx_host_alloc(size_t bytes, void** out);
int ffx_host_free(void* p);
int ffx_host_pin(void* p, size_t bytes);
int ffx_host_unpin(void* p);

#ifdef __cplusplus
}
#endif
#endif /* FFX_H_ */
