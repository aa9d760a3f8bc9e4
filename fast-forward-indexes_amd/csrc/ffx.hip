// ffx.hip -- C ABI (include/ffx.h) over the HIP kernels.  gfx950 only.
//
// Host side of the re-ranking path: device memory for the index and for one batch of
// queries, the launch sequence  count -> scan -> expand -> gather_dot -> reduce [-> sort],
// hipEvent timing of each stage, and status-code error reporting (no exceptions cross the ABI).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <functional>
#include <future>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>  // before rocprim: its texture iterator calls memset unqualified

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <atomic>
#include <memory>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/ffx.h"
#include "../../include/ffx_private.h"
#include "ffx_internal.h"
#include "host_xfer.h"
#include "aux_kernels.hip.h"
#include "gather_dot.hip.h"

thread_local std::string g_last_error;

// FFX_TRACE_HOST=1 (a diagnostic, off by default): one line on stderr per operation that hands host addresses to the runtime
// outside the scoring paths -- page-locked allocations, device pulls, row gathers -- so that the address of a device fault
// ("Memory access fault by GPU node ... on address ...", the runtime's last words) can be matched with who owned it.
static bool ffx_trace_host() {
  static const bool on = [] {
    const char* e = getenv("FFX_TRACE_HOST");
    return e && *e && *e != '0';
  }();
  return on;
}

// FFX_ABORT_BACKTRACE=1 (a diagnostic, off by default): when something below this library -- the HIP runtime, the driver's
// user-mode part -- ends the process with abort(), the native call stack of the aborting thread goes to stderr first (Python's
// faulthandler shows the Python frames only).  For the rare SIGABRT inside hipMemcpy of HISTORY.md section 10 / tools/abort_hunt.sh.
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
namespace {
void ffx_abort_backtrace(int) {
  static const char head[] = "\nlibffx: SIGABRT -- native call stack of the aborting thread:\n";
  (void)!write(2, head, sizeof head - 1);
  void* frames[64];
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  signal(SIGABRT, SIG_DFL);
  raise(SIGABRT);
}
// (Rounds 3 - 4 lowered the runtime's on-the-fly page-locking threshold from a constructor here, GPU_PINNED_MIN_XFER_SIZE=32: a
// process-wide setting changed behind the host program's back.  Gone: every transfer between caller memory and the device now
// goes through host_xfer.h, which stages ordinary memory through page-locked blocks of the library's own, so the runtime is
// never handed pages it would lock -- whatever its threshold, and whether or not HIP was initialised before this library loaded.)
__attribute__((constructor)) void ffx_install_abort_backtrace() {
  const char* e = getenv("FFX_ABORT_BACKTRACE");
  if (e && *e && *e != '0') {
    void* warm[2];
    (void)backtrace(warm, 2);  // (loads libgcc now: not inside the handler)
    signal(SIGABRT, ffx_abort_backtrace);
  }
}
}  // namespace

namespace {

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

}  // namespace

extern "C" int ffx_build_has_experiments(void) {
#ifdef FFX_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

// shared with host_io.cpp
int ffx_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

// FFX_DEBUG_VARIANT -- the ONE environment knob for A/B switches (tests, tools/*.py; never needed in production).
// A comma-separated list of words, `name` or `name=N`.  In every build (a choice between two paths that both ship):
//   no_small_path          ffx_score never takes the single-submission path of small batches
//   route_host             the sharded scorer routes a batch on the host even for document-range shards
//   pq_no_fuse, pq_no_adc, pq_sort_apart  product-quantized rows: unfused exact ADC / decode-then-dot instead of exact ADC / the ordering as a launch of its own
//   no_derive              d = 768 / 1024: the row list written by an expand kernel instead of rows derived inside gather_dot
//   score_slice_pairs=N, score_blocking_out   where ffx_score cuts a batch into slices; results fetched by the calling thread
//   stage_bytes=N          size of a page-locked staging block of ffx_index_upload_rows
//   rehearsal_paced[=GB/s] the rehearsal transport's stand-in exchange paced like a link (bench.py --rehearsal-exchange)
// Only in the experiments build (make EXPERIMENTS=1; ffx_experiment in ffx_internal.h -- tuning hooks that lost their A/B):
//   gather=N               a gather_dot instantiation by variant number (tools/kernel_ab.py): 8xx strided small rows, 9xx d = 768 forms,
//                          921 - 924 the persistent grid, 925 workgroups of two waves
//   generic_staged, generic_plain, anyd_u=N   the round-1 generic exact kernels instead of the any-d / segment kernels; loads in flight of the any-d kernel
//   sort_device_wide_from=N, sort_block_from=N, sort_block_to=N, sort_interp_apart   where the per-query sort switches form; the interpolation as a launch of its own
//   score_plans=N, score_ramp_pairs=N, score_ramp_growth_pct=N, score_no_ramp, score_no_out_stream, score_out_priority_low, pull_grid=N, count_always
//                          the slicing of ffx_score's host-to-host pipeline (tools/score_trace.py, slice_sweep.py)
//   route_high_priority, merge_on_results, merge_on_compute   stream placement of the sharded scorer's routing and merge
// (FFX_SCORE_TRACE=1 -- a timeline on stderr -- and the operational settings FFX_COMM_TIMEOUT_S, FFX_RDZV_DIR,
// FFX_RDZV_KEY, FFX_READ_THREADS are not variants and keep their own names.)
// True if `name` is in the list; *value (if given) receives N, or stays untouched for a bare word.
bool ffx_debug_variant(const char* name, long long* value) {
  const char* list = getenv("FFX_DEBUG_VARIANT");
  if (!list || !*list) return false;
  const size_t len = strlen(name);
  for (const char* p = list; *p;) {
    const char* end = strchr(p, ',');
    const size_t n = end ? (size_t)(end - p) : strlen(p);
    if (n >= len && strncmp(p, name, len) == 0 && (n == len || p[len] == '=')) {
      if (value && n > len) *value = atoll(p + len + 1);
      return true;
    }
    if (!end) break;
    p = end + 1;
  }
  return false;
}

namespace {

#define FFX_HIP(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(e_ == hipErrorOutOfMemory ? FFX_OOM : FFX_HIP_ERROR, "%s: %s (%s:%d)", #expr, \
                  hipGetErrorString(e_), __FILE__, __LINE__);                                 \
  } while (0)

#define FFX_TRY(expr)          \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != FFX_OK) return rc_; \
  } while (0)

// growable device buffer
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {
    if (bytes <= cap) return FFX_OK;
    if (p) {
      FFX_HIP(hipFree(p));
      p = nullptr;
      cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      want = bytes;
      e = hipMalloc(&p, want);
    }
    if (e != hipSuccess) {
      p = nullptr;
      return fail(FFX_OOM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    }
    cap = want;
    return FFX_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const { return static_cast<T*>(p); }
};

inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// leaf boundaries of NumPy's pairwise recursion for a row of n elements (in-order)
void numpy_leaf_plan(int start, int n, std::vector<int>& bounds) {
  if (n <= 128) {
    bounds.push_back(start);
    return;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  numpy_leaf_plan(start, n2, bounds);
  numpy_leaf_plan(start + n2, n - n2, bounds);
}

// The order in which NumPy's recursion adds the leaf sums, as a list of (a, b) steps
// "sum[a] += sum[b]" over the in-order leaf indices; the total ends up in sum[0].
// A long row cut into segments of at most 8 leaves each: NumPy's recursion is followed from the top until a piece
// has <= 8 leaves (its sub-tree then is what gather_dot_exact_anyd_kernel sums); `steps` replays the recursion above
// the pieces ("sum[a] += sum[b]" over segment numbers, the total ending in segment 0).  Returns the number of the
// segment that holds the sum of [start, start + n).
int g_max_piece_leaves = 8;  // (experiments: smaller pieces, see plan_create)
int long_row_plan(int start, int n, std::vector<std::pair<int, int>>& pieces, std::vector<int>& steps) {
  std::vector<int> bounds;
  numpy_leaf_plan(0, n, bounds);
  if ((int)bounds.size() <= g_max_piece_leaves) {
    pieces.emplace_back(start, n);
    return (int)pieces.size() - 1;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  const int a = long_row_plan(start, n2, pieces, steps);
  const int b = long_row_plan(start + n2, n - n2, pieces, steps);
  steps.push_back(a);
  steps.push_back(b);
  return a;
}

int numpy_combine_plan(int n, int& next_leaf, std::vector<int>& steps) {
  if (n <= 128) return next_leaf++;
  int n2 = n / 2;
  n2 -= n2 % 8;
  const int a = numpy_combine_plan(n2, next_leaf, steps);
  const int b = numpy_combine_plan(n - n2, next_leaf, steps);
  steps.push_back(a);
  steps.push_back(b);
  return a;
}

}  // namespace

struct ffx_index {
  int device = 0;
  int32_t dim = 0;
  int32_t dtype = FFX_DTYPE_F32;
  size_t elem = sizeof(float);  // bytes per stored element
  int64_t capacity = 0;
  void* vectors = nullptr;
  char* row_ptr(int64_t row) const { return static_cast<char*>(vectors) + (size_t)row * dim * elem; }
  // what a query vector looks like: dim for float/half rows, M*Ds for product-quantized rows
  int32_t qdim = 0;
  DevBuf codebook;    // FFX_DTYPE_PQ8: float[M][Ks][Ds]
  DevBuf codebook_t;  // the same, transposed to [M][Ds][Ks] (coalesced reads when tabulating ADC)
  int32_t pq_ks = 0, pq_ds = 0;
  // id structure (device copies)
  DevBuf doc_off;   // int32[n_docs+1]
  DevBuf doc_rows;  // int32[n_doc_rows]
  DevBuf psg_row;   // int32[n_psgs]
  DevBuf doc_first; // int32[n_docs]: first row of every document (FIRSTP / single-passage docs)
  int64_t n_docs = 0, n_doc_rows = 0, n_psgs = 0;
  int64_t max_doc_rows = 1;  // rows of the longest document: bounds what a candidate can contribute
  bool docs_identity = false;  // document ordinal o owns exactly row o
  bool doc_rows_identity = false;  // doc_rows[i] == i: documents were added whole and in order (position in doc_rows == row)
  bool psgs_identity = false;  // passage ordinal o is row o
  bool docs_nonempty = false;  // every document has at least one row
  // the plans ffx_score rotates through (two; three when results leave asynchronously); created on first use, freed with the index
  static constexpr int kScorePlans = 5;
  struct ffx_plan* score_plans[kScorePlans] = {};
  // page-locked staging of ffx_index_upload_rows: two blocks, filled by host threads while the
  // other one is in flight to the device
  void* stage[2] = {nullptr, nullptr};
  size_t stage_bytes = 0;
  hipEvent_t stage_ev[2] = {nullptr, nullptr};
  bool stage_busy[2] = {false, false};  // a copy out of the block may still be in flight (ffx_index_upload_rows_async)
  int64_t pending_row = 0, pending_rows = 0;  // rows whose codes ffx_index_upload_wait still has to check (PQ rows)
  // ffx_index_pull_rows_async: the row list and the chunk maps' device addresses (kept while the maps stay the same)
  DevBuf pull_list, pull_chunks;
  std::vector<const void*> pull_chunk_hosts;
  int64_t* pull_list_host = nullptr;  // page-locked, pull_list_room entries
  hipEvent_t order_ev = nullptr;      // ffx_index_upload_after
  int64_t pull_list_room = 0;
  hipStream_t copy_stream = nullptr;
  // ffx_score with page-locked result buffers: results leave on this stream (highest priority, so that the
  // copy of slice c is not queued behind the scoring kernels of slice c+1), one event per plan
  hipStream_t out_stream = nullptr;
  hipStream_t score_stream = nullptr;  // ... and every slice's kernels run on this one, in order (see ffx_score)
  hipEvent_t out_done[kScorePlans] = {};
  // FFX_FLAG_KEEP_DENSE: the dense scores (first-stage order) and the carried words of the last such ffx_score_ranked call,
  // for ffx_rank_kept (Ranking.interpolate on an Index.__call__ result: the interpolation and the ordering, no gather)
  void* gather_stage = nullptr;  // ffx_index_gather's page-locked block (rows never go straight into the caller's memory)
  size_t gather_stage_bytes = 0;
  DevBuf kept_dense, kept_carry;
  int64_t kept_pairs = -1, kept_queries = -1;
  bool kept_has_carry = false;
  uint64_t kept_token = 0, kept_off_hash = 0;
};

struct ffx_plan {
  ffx_index* idx = nullptr;
  int device = 0;                    // idx->device, kept so that destroying the plan never touches the index
  hipStream_t stream = nullptr;      // the stream work is issued on
  hipStream_t own_stream = nullptr;  // created with the plan; `stream` may point elsewhere
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // gather_dot's start / end of the most recent runs, so that a caller who queues runs back to back (no wait in
  // between) can still read every run's kernel time afterwards (ffx_plan_recent_gather_ms); created on first use
  static constexpr int kRecent = 256;
  hipEvent_t recent[kRecent][2] = {};
  uint64_t n_runs = 0;
  // inputs
  DevBuf queries, pair_off, pair_ids, sparse;
  // workspace
  DevBuf cnt, seg_off, desc, rs, scan_tmp, sort_tmp, iota, sorted_keys, leaf_start, flags, lut, vdesc, rs_v, combine_steps;
  // set while a long row is scored as S short ones (launch_gather_dot_t): the row length, row
  // count and row list the kernels are to see
  int ov_dim = 0, ov_rows = 0;
  // outputs
  DevBuf ff, interp, order;
  // the run in final order (ffx_score_ranked): the sort key and a carried word per candidate, permuted by `order`
  DevBuf carry, ranked, carry_ranked;
  // documents of several passages scored without a row list (gather_dot.hip.h, Derive): made on the first run of a candidate list
  DevBuf pair_first, tile_tab;
  bool derive_valid = false, use_derive = false, use_derive1 = false;
  // for the one-row derive: the query of every 1024th candidate (made on the host from the offsets it validates anyway and
  // uploaded with them), so that a wave finds its tile's query in a step or two however ragged the lists are
  static constexpr int64_t kHintBlock = int64_t(1) << ffx::kQueryHintShift;
  DevBuf q_hint;
  std::vector<int32_t> host_hint;
  bool q_hint_valid = false;
  bool forced_pieces = false;  // (experiments: seg_leaves=N)
  DevBuf dm_keys, dm_keys2, dm_perm, dm_perm2, dm_desc, dm_rs, dm_tmp;  // (experiments: doc_major)
  bool want_ranked = false, have_carry = false;
  bool dense_given = false;  // the next run starts from dense scores already in `ff` (ffx_rank_kept): no expand, no gather
  int64_t n_queries = 0, n_pairs = 0, n_rows = 0;
  int n_queries_set = 0;
  int mode = 0;
  bool have_sparse = false, have_pairs = false, ran = false, ran_sort = false;
  int n_leaves = 0;
  bool anyd_ok = false, anyd_tails = false;  // every leaf has 8..128 elements; some leaf has len % 8 != 0
  // a row of more than 8 leaves, cut along NumPy's recursion into segments of at most 8 leaves each (long_row_plan):
  // per segment where it starts, how long it is, and where its leaf table / combine steps lie in seg_tables
  struct Segment {
    int col0, len, n_leaves, table_at, steps_at;
    bool tails;
  };
  std::vector<Segment> segments;
  DevBuf seg_tables;   // int: per segment its leaf bounds (n_leaves + 1, relative to the segment) then its combine steps
  int seg_steps_at = 0, seg_n_steps = 0;  // the steps that add the segment sums (in seg_tables)
  bool last_interp = false;
  int32_t variant = 0;
  int32_t order_base = 0;  // added to the positions the per-query order is made of (slices of a larger batch)
  std::vector<int64_t> host_pair_off;
  // early stopping (ffx_plan_early_stopping): the interval's own CSR lives in a child plan that
  // shares this plan's stream; pairs set "lazily" have not been validated yet
  struct ffx_plan* es_child = nullptr;
  DevBuf es_active, es_cnt, es_pos, es_scan_tmp;
  DevBuf small_flag;  // ffx_score's single-submission path: smallest bad pair position, ~0 between calls
  bool lazy_pairs = false;
  // the ordinal check of set_pairs between its launch and its completion: where the verdict lands on the host
  bool check_one_each = false;
  char* check_back = nullptr;
  char check_host[16] = {};
  bool adopt_empty_index = false;
  // Small batches (a service scoring one query at a time) are latency-bound on the copies: a
  // copy between pageable memory and the device blocks the caller each time, so inputs and
  // outputs of up to kStageBytes travel through this page-locked block and only one wait remains.
  static constexpr size_t kStageBytes = 4 << 20;
  // results of a slice of ffx_score on their way into ORDINARY memory of the caller (the helper thread of that slice): two
  // page-locked blocks and a stream of the plan's own, see staged_d2h
  static constexpr size_t kOutStageBytes = 8 << 20;  // (a copy command costs the wire ~20 us whatever its size: 4 MiB pieces leave at 45 GB/s, 8 MiB at 51)
  static constexpr int kOutLanes = 10;  // two per result array of a slice (dense, interpolated, order, ranked, carried): they leave side by side, in halves
  char* out_stage[kOutLanes][2] = {};
  hipEvent_t out_stage_ev[kOutLanes][2] = {};
  char* hstage = nullptr;
  size_t hstage_used = 0;
  // room for `bytes` more in the staging block (nullptr: too large, or no page-locked memory)
  char* stage(size_t bytes) {
    if (!hstage) {
      if (hstage_failed || hipHostMalloc(reinterpret_cast<void**>(&hstage), kStageBytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        hstage = nullptr;
        hstage_failed = true;
        return nullptr;
      }
    }
    const size_t at = (hstage_used + 63) & ~size_t(63);
    if (at + bytes > kStageBytes) return nullptr;
    hstage_used = at + bytes;
    return hstage + at;
  }
  bool hstage_failed = false;
};

static bool derive_applies(const ffx_plan* pl);  // (both defined beside the gather_dot dispatcher)
static bool derive_kernel_exists(const ffx_plan* pl);

namespace {
// host -> device on the plan's stream; small sources go through the page-locked block (the
// caller's buffer is free again on return either way)
int plan_h2d(ffx_plan* pl, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return FFX_OK;
  if (char* st = bytes <= (1u << 20) ? pl->stage(bytes) : nullptr) {
    memcpy(st, src, bytes);
    src = st;
  }
  FFX_TRY(ffx_h2d(pl->device, dst, src, bytes, pl->stream));
  return FFX_OK;
}
}  // namespace

namespace {
// A product-quantized index whose quantizer has fewer than 256 codewords per subspace: code bytes
// >= Ks (a corrupt or foreign index file) would index past the codebook and the ADC tables.  The
// kernels clamp; uploads refuse.  Checks device rows [row, row + n_rows) after they have landed.
int check_pq_codes(ffx_index* idx, int64_t row, int64_t n_rows) {
  if (idx->dtype != FFX_DTYPE_PQ8 || idx->pq_ks <= 0 || idx->pq_ks >= 256 || n_rows <= 0) return FFX_OK;
  unsigned long long* flag = nullptr;
  FFX_HIP(hipMalloc(reinterpret_cast<void**>(&flag), sizeof *flag));
  unsigned long long bad = ~0ull;
  const long n_bytes = (long)n_rows * idx->dim;
  hipError_t e = hipMemset(flag, 0xFF, sizeof *flag);
  if (e == hipSuccess) {
    ffx::pq_check_codes_kernel<<<blocks_for(n_bytes, 256), 256>>>(reinterpret_cast<const uint8_t*>(idx->row_ptr(row)), n_bytes, idx->pq_ks, flag);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(&bad, flag, sizeof bad, hipMemcpyDeviceToHost);  // (8 bytes: the runtime's own staging)
  (void)hipFree(flag);
  if (e != hipSuccess) return fail(FFX_HIP_ERROR, "code check failed: %s", hipGetErrorString(e));
  if (bad != ~0ull)
    return fail(FFX_BAD_ARG, "row %lld holds a code >= Ks = %d in subspace %lld: not codes of this index's quantizer",
                (long long)(row + (int64_t)(bad / (unsigned long long)idx->dim)), idx->pq_ks, (long long)(bad % (unsigned long long)idx->dim));
  return FFX_OK;
}
}  // namespace

// --------------------------------------------------------------------------------------------
extern "C" {

int ffx_abi_version(void) { return FFX_ABI_VERSION; }

const char* ffx_last_error(void) { return g_last_error.c_str(); }

int ffx_device_count(int* count) {
  if (!count) return fail(FFX_BAD_ARG, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return FFX_OK;
}

int ffx_device_info(int device, char* name, size_t name_cap, int64_t* total_bytes, int64_t* free_bytes,
                    int32_t* compute_units) {
  hipDeviceProp_t prop;
  FFX_HIP(hipGetDeviceProperties(&prop, device));
  if (name && name_cap) {
    snprintf(name, name_cap, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (total_bytes || free_bytes) {
    FFX_HIP(hipSetDevice(device));
    size_t f = 0, t = 0;
    FFX_HIP(hipMemGetInfo(&f, &t));
    if (total_bytes) *total_bytes = (int64_t)t;
    if (free_bytes) *free_bytes = (int64_t)f;
  }
  return FFX_OK;
}

// ---- index ------------------------------------------------------------------------------------
int ffx_index_create(int device, int64_t capacity_rows, int32_t dim, ffx_index** out) {
  return ffx_index_create_typed(device, capacity_rows, dim, FFX_DTYPE_F32, out);
}

int ffx_index_dtype(const ffx_index* idx, int32_t* dtype) {
  if (!idx || !dtype) return fail(FFX_BAD_ARG, "NULL argument");
  *dtype = idx->dtype;
  return FFX_OK;
}

int ffx_index_create_typed(int device, int64_t capacity_rows, int32_t dim, int32_t dtype, ffx_index** out) {
  if (!out) return fail(FFX_BAD_ARG, "out is NULL");
  *out = nullptr;
  if (dtype != FFX_DTYPE_F32 && dtype != FFX_DTYPE_F16 && dtype != FFX_DTYPE_PQ8) return fail(FFX_BAD_ARG, "unknown dtype %d", dtype);
  if (capacity_rows < 0 || capacity_rows > INT32_MAX) return fail(FFX_BAD_ARG, "capacity_rows out of range (row ids are int32)");
  if (dim <= 0) return fail(FFX_BAD_ARG, "dim must be positive");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    (void)hipGetLastError();
    return fail(FFX_NO_DEVICE, "no HIP device: libffx has no CPU fallback");
  }
  if (device < 0 || device >= n) return fail(FFX_BAD_ARG, "device %d out of range (%d devices)", device, n);
  FFX_HIP(hipSetDevice(device));
  ffx_index* idx = new (std::nothrow) ffx_index();
  if (!idx) return fail(FFX_OOM, "host allocation failed");
  idx->device = device;
  idx->dim = dim;
  idx->dtype = dtype;
  idx->elem = dtype == FFX_DTYPE_F16 ? 2 : (dtype == FFX_DTYPE_PQ8 ? 1 : 4);
  idx->qdim = dtype == FFX_DTYPE_PQ8 ? 0 : dim;  // PQ: known once the codebook is set
  idx->capacity = capacity_rows;
  const size_t bytes = std::max<size_t>((size_t)capacity_rows * dim * idx->elem, 256);
  hipError_t e = hipMalloc(&idx->vectors, bytes);
  if (e != hipSuccess) {
    delete idx;
    return fail(FFX_OOM, "hipMalloc of %zu index bytes failed: %s", bytes, hipGetErrorString(e));
  }
  *out = idx;
  return FFX_OK;
}

int ffx_index_destroy(ffx_index* idx) {
  if (!idx) return FFX_OK;
  (void)hipSetDevice(idx->device);
  for (auto& pl : idx->score_plans) {
    if (pl) ffx_plan_destroy(pl);
    pl = nullptr;
  }
  for (int i = 0; i < 2; ++i) {
    if (idx->stage[i]) (void)hipHostFree(idx->stage[i]);
    if (idx->stage_ev[i]) (void)hipEventDestroy(idx->stage_ev[i]);
  }
  if (idx->copy_stream) (void)hipStreamDestroy(idx->copy_stream);
  if (idx->pull_list_host) (void)hipHostFree(idx->pull_list_host);
  if (idx->order_ev) (void)hipEventDestroy(idx->order_ev);
  idx->pull_list.release();
  idx->pull_chunks.release();
  idx->kept_dense.release();
  idx->kept_carry.release();
  if (idx->gather_stage) (void)hipHostFree(idx->gather_stage);
  if (idx->out_stream) (void)hipStreamDestroy(idx->out_stream);
  if (idx->score_stream) (void)hipStreamDestroy(idx->score_stream);
  for (auto& e : idx->out_done)
    if (e) (void)hipEventDestroy(e);
  if (idx->vectors) (void)hipFree(idx->vectors);
  idx->doc_off.release();
  idx->doc_rows.release();
  idx->psg_row.release();
  idx->doc_first.release();
  idx->codebook.release();
  idx->codebook_t.release();
  delete idx;
  return FFX_OK;
}

int ffx_index_dim(const ffx_index* idx, int32_t* dim) {
  if (!idx || !dim) return fail(FFX_BAD_ARG, "NULL argument");
  *dim = idx->dim;
  return FFX_OK;
}

int ffx_index_query_dim(const ffx_index* idx, int32_t* dim) {
  if (!idx || !dim) return fail(FFX_BAD_ARG, "NULL argument");
  *dim = idx->qdim;
  return FFX_OK;
}

int ffx_index_set_codebook(ffx_index* idx, const float* codewords, int32_t m, int32_t ks, int32_t ds) {
  if (!idx || !codewords) return fail(FFX_BAD_ARG, "NULL argument");
  if (idx->dtype != FFX_DTYPE_PQ8) return fail(FFX_BAD_ARG, "the index does not hold product-quantizer codes");
  if (m != idx->dim) return fail(FFX_BAD_ARG, "codebook has %d subspaces, rows have %d codes", m, idx->dim);
  if (ks < 1 || ks > 256 || ds < 1) return fail(FFX_BAD_ARG, "bad codebook shape (Ks must be <= 256 for uint8 codes)");
  if ((int64_t)m * ds > (1 << 20)) return fail(FFX_BAD_ARG, "decoded dimensionality too large");
  FFX_HIP(hipSetDevice(idx->device));
  for (auto& pl : idx->score_plans) {  // their leaf plan follows the decoded dimensionality
    if (pl) ffx_plan_destroy(pl);
    pl = nullptr;
  }
  const size_t bytes = (size_t)m * ks * ds * sizeof(float);
  FFX_TRY(idx->codebook.ensure(bytes));
  FFX_TRY(ffx_h2d_now(idx->device, idx->codebook.p, codewords, bytes));
  std::vector<float> tr((size_t)m * ks * ds);
  for (int a = 0; a < m; ++a)
    for (int b = 0; b < ks; ++b)
      for (int c = 0; c < ds; ++c) tr[((size_t)a * ds + c) * ks + b] = codewords[((size_t)a * ks + b) * ds + c];
  FFX_TRY(idx->codebook_t.ensure(bytes));
  FFX_TRY(ffx_h2d_now(idx->device, idx->codebook_t.p, tr.data(), bytes));
  idx->pq_ks = ks;
  idx->pq_ds = ds;
  idx->qdim = m * ds;
  return FFX_OK;
}

int ffx_index_device_ptr(const ffx_index* idx, void** ptr) {
  if (!idx || !ptr) return fail(FFX_BAD_ARG, "NULL argument");
  *ptr = idx->vectors;
  return FFX_OK;
}

// the placement probe over one block of the index's size: time of a random gather of whole rows (best of three, after one uncounted launch)
static int probe_placement(const ffx_index* idx, const void* block, unsigned* sink, hipEvent_t* ev, float* out_ms) {
  const size_t row_bytes = (size_t)idx->dim * idx->elem;
  const size_t bytes = std::max<size_t>((size_t)idx->capacity * row_bytes, 256);
  // rows as the scoring kernels read them (16-byte loads); rows that are no multiple of 16 bytes (PQ codes): 128-byte lines
  const bool whole = row_bytes >= 16 && row_bytes % 16 == 0;
  const int row_vec = whole ? (int)(row_bytes / 16) : 8;
  const long n_rows = whole ? (long)idx->capacity : (long)(bytes / 128);
  if (n_rows < 1) return fail(FFX_BAD_ARG, "an empty index has no placement to probe");
  long n_probe = (long)std::min<size_t>(size_t(1) << 23, (size_t(32) << 30) / ((size_t)row_vec * 16));  // (8 M rows: ~4 ms at 3 KB a row)
  n_probe = std::max<long>(8, n_probe / 8 * 8);
  // launches that are not counted come first, 25 ms of them: the translation caches' upper levels fill, and a device that sat idle
  // (the settling loop of ffx_index_reseat sleeps between its looks) is back at its clocks -- a 2 ms probe right after a pause reads
  // 15 % slow otherwise; then the best of three
  float best = 0.f, warm = 0.f, last = 0.f;
  int counted = 0;
  for (int launch = 0; counted < 3 && launch < 64; ++launch) {
    FFX_HIP(hipEventRecord(ev[0], nullptr));
    ffx::placement_probe_kernel<<<blocks_for(n_probe, 32), 256, 0, nullptr>>>(static_cast<const ffx::u32x4*>(block), n_rows, row_vec, n_probe,
                                                                             0x5eedull + launch, sink);
    FFX_HIP(hipEventRecord(ev[1], nullptr));
    FFX_HIP(hipEventSynchronize(ev[1]));
    float ms = 0.f;
    FFX_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
    last = ms;
    if (warm < 25.f) {
      warm += ms;
      continue;
    }
    if (counted == 0 || ms < best) best = ms;
    ++counted;
  }
  *out_ms = counted ? best : last;  // (launches too short to add up to the warm-up: the last one stands)
  return FFX_OK;
}

// ffx_private.h: the probe alone, over the block the index has (tools/placement_probe.py compares it with the scoring kernel)
int ffx_index_probe_placement(ffx_index* idx, float* ms) {
  if (!idx || !ms) return fail(FFX_BAD_ARG, "NULL argument");
  FFX_HIP(hipSetDevice(idx->device));
  FFX_HIP(hipDeviceSynchronize());
  DevBuf sink;
  FFX_TRY(sink.ensure(65536 * sizeof(unsigned)));
  hipEvent_t ev[2] = {nullptr, nullptr};
  FFX_HIP(hipEventCreate(&ev[0]));
  hipError_t e = hipEventCreate(&ev[1]);
  int rc = e == hipSuccess ? probe_placement(idx, idx->vectors, sink.as<unsigned>(), ev, ms) : fail(FFX_HIP_ERROR, "hipEventCreate: %s", hipGetErrorString(e));
  for (int i = 0; i < 2; ++i)
    if (ev[i]) (void)hipEventDestroy(ev[i]);
  return rc;
}

// ffx.h: the rows move into the fastest of a few allocations.  How fast a random gather over 27 GB runs belongs to the
// ALLOCATION: eight copies of the headline index side by side in one process score the same batch in 15.95 .. 16.92 ms, each
// within 0.3 % of its own figure round after round (tools/placement_probe.py; the translation counters of HISTORY section 5
// show two regimes of the L1 translation cache).  Nothing the library does at hipMalloc selects the memory it is handed; what
// it can do is look at a few hands.
int ffx_index_reseat(ffx_index* idx, int32_t tries, ffx_reseat_report* report) {
  if (!idx) return fail(FFX_BAD_ARG, "NULL argument");
  if (tries < -1 || tries > 7) return fail(FFX_BAD_ARG, "tries must be 0 .. 7, or -1 for as many as half of the free memory holds");
  FFX_HIP(hipSetDevice(idx->device));
  const size_t row_bytes = (size_t)idx->dim * idx->elem;
  const size_t bytes = std::max<size_t>((size_t)idx->capacity * row_bytes, 256);
  if (tries < 0) {  // as many as fit into half of what is free now (never all of it: other users of the device allocate too)
    size_t free_b = 0, total_b = 0;
    FFX_HIP(hipMemGetInfo(&free_b, &total_b));
    tries = (int32_t)std::min<size_t>(7, free_b / 2 / bytes);
  }
  ffx_reseat_report rep{};
  rep.bytes = (int64_t)bytes;
  rep.candidates = 1;
  if (tries == 0 || bytes < (size_t(256) << 20)) {  // (a small index lies within the reach of the translation caches wherever it is)
    if (report) *report = rep;
    return FFX_OK;
  }
  FFX_HIP(hipDeviceSynchronize());  // nothing of this index may be in flight
  std::vector<void*> blocks{idx->vectors};
  hipEvent_t ev[2] = {nullptr, nullptr};
  DevBuf sink;
  struct Guard {  // whatever happens below: every block but the index's own goes back
    std::vector<void*>& blocks;
    ffx_index* idx;
    hipEvent_t* ev;
    ~Guard() {
      for (void* p : blocks)  // (an error path: the normal one has freed them itself)
        if (p && p != idx->vectors) (void)hipFree(p);
      for (int i = 0; i < 2; ++i)
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    }
  } guard{blocks, idx, ev};
  for (int t = 0; t < tries; ++t) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();  // memory ran out: the candidates so far are the candidates
      break;
    }
    blocks.push_back(p);
  }
  FFX_TRY(sink.ensure(65536 * sizeof(unsigned)));
  FFX_HIP(hipEventCreate(&ev[0]));
  FFX_HIP(hipEventCreate(&ev[1]));
  // every candidate gets the rows BEFORE it is timed: the probe over memory nobody has written reads faster than over the same
  // block with the rows in it (13.6 GB of float16 rows: 1.88 ms against 2.17), so empty candidates would always beat the block in use
  for (size_t i = 1; i < blocks.size(); ++i) FFX_HIP(hipMemcpyAsync(blocks[i], idx->vectors, bytes, hipMemcpyDeviceToDevice, nullptr));
  FFX_HIP(hipStreamSynchronize(nullptr));
  for (size_t i = 0; i < blocks.size(); ++i) FFX_TRY(probe_placement(idx, blocks[i], sink.as<unsigned>(), ev, &rep.probe_ms[i]));
  rep.candidates = (int32_t)blocks.size();
  int chosen = 0;
  for (int i = 1; i < rep.candidates; ++i)
    if (rep.probe_ms[i] < rep.probe_ms[chosen]) chosen = i;
  if (chosen != 0) idx->vectors = blocks[(size_t)chosen];  // (it holds the rows already; what the index does not hold is freed below)
  rep.chosen = chosen;
  // The losers go back now -- and the device is not at rest when they have: for a few seconds after blocks of this size are freed,
  // gathers run 1 - 3 % slower (tools/reseat_check.py: 16.76 ms right after, 16.33 from four seconds on; what the driver does with
  // freed memory is its business).  The call returns when that spell is over.
  bool freed = false;
  for (void*& p : blocks)
    if (p && p != idx->vectors) {
      (void)hipFree(p);
      p = nullptr;
      freed = true;
    }
  // (every run of tools/reseat_check.py saw the slow spell end four seconds after the frees, to the tenth of a second -- a timer of
  // the driver's, it seems -- and the probe does not always see the spell: 3.90 ms against 3.89 while the step ran 16.3 instead of
  // 16.1 ms.  So: not before 4.25 s have passed, then until the probe agrees, at most 8 s in all)
  const auto t0 = std::chrono::steady_clock::now();
  const float target = rep.probe_ms[chosen] * 1.01f;
  if (freed) std::this_thread::sleep_for(std::chrono::milliseconds(4250));
  for (;;) {
    FFX_TRY(probe_placement(idx, idx->vectors, sink.as<unsigned>(), ev, &rep.probe_ms_settled));
    rep.settle_s = std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count();
    if (rep.probe_ms_settled <= target || rep.settle_s > 8.f) break;
    std::this_thread::sleep_for(std::chrono::milliseconds(250));
  }
  if (report) *report = rep;
  return FFX_OK;
}

int ffx_index_capacity(const ffx_index* idx, int64_t* rows) {
  if (!idx || !rows) return fail(FFX_BAD_ARG, "NULL argument");
  *rows = idx->capacity;
  return FFX_OK;
}

int ffx_index_upload(ffx_index* idx, const void* host_rows, int64_t row_offset, int64_t n_rows) {
  if (!idx || (!host_rows && n_rows > 0)) return fail(FFX_BAD_ARG, "NULL argument");
  if (row_offset < 0 || n_rows < 0 || row_offset + n_rows > idx->capacity)
    return fail(FFX_BAD_ARG, "rows [%lld, %lld) outside capacity %lld", (long long)row_offset,
                (long long)(row_offset + n_rows), (long long)idx->capacity);
  if (n_rows == 0) return FFX_OK;
  FFX_HIP(hipSetDevice(idx->device));
  FFX_TRY(ffx_h2d_now(idx->device, idx->row_ptr(row_offset), host_rows, (size_t)n_rows * idx->dim * idx->elem));
  return check_pq_codes(idx, row_offset, n_rows);
}

// Rows scattered over the chunks of a memory-mapped file -> consecutive device rows.  Host threads
// gather one block of rows into page-locked staging while the previous block travels to the
// device (the streaming half of OnDiskIndex: disk.py:100-121 reads the same chunk maps).
int ffx_index_upload_rows(ffx_index* idx, const void* const* chunks, int64_t n_chunks, int64_t chunk_rows,
                          const int64_t* rows, int64_t n_rows, int64_t dst_row, int32_t n_threads) {
  FFX_TRY(ffx_index_upload_rows_async(idx, chunks, n_chunks, chunk_rows, rows, n_rows, dst_row, n_threads));
  return ffx_index_upload_wait(idx);
}

// The same without waiting for the last blocks to land: returns once every row has been gathered and its copy queued,
// so that the caller's next gather (another index's rows, e.g. the next window of a streamed index) overlaps the tail
// of this one's copies.  ffx_index_upload_wait(idx) before the rows are used.
int ffx_index_upload_rows_async(ffx_index* idx, const void* const* chunks, int64_t n_chunks, int64_t chunk_rows,
                                const int64_t* rows, int64_t n_rows, int64_t dst_row, int32_t n_threads) {
  if (!idx || n_rows < 0 || dst_row < 0 || n_chunks < 0 || chunk_rows <= 0) return fail(FFX_BAD_ARG, "bad argument");
  if (n_rows == 0) return FFX_OK;
  if (!chunks || !rows) return fail(FFX_BAD_ARG, "NULL argument");
  if (dst_row + n_rows > idx->capacity) return fail(FFX_BAD_ARG, "rows [%lld, %lld) exceed the capacity %lld", (long long)dst_row, (long long)(dst_row + n_rows), (long long)idx->capacity);
  const int64_t n_src = n_chunks * chunk_rows;
  for (int64_t i = 0; i < n_rows; ++i)
    if (rows[i] < 0 || rows[i] >= n_src) return fail(FFX_BAD_ARG, "rows[%lld] = %lld is outside the %lld stored rows", (long long)i, (long long)rows[i], (long long)n_src);
  for (int64_t c = 0; c < n_chunks; ++c)
    if (!chunks[c]) return fail(FFX_BAD_ARG, "chunk %lld is NULL", (long long)c);
  FFX_HIP(hipSetDevice(idx->device));
  const size_t rb = (size_t)idx->dim * idx->elem;
  size_t want = (size_t)64 << 20;
  if (long long v = 0; ffx_debug_variant("stage_bytes", &v)) want = std::max<size_t>(rb, (size_t)v);
  want = std::max(want, rb);
  if (idx->pending_rows > 0) FFX_TRY(ffx_index_upload_wait(idx));  // (an earlier asynchronous upload nobody waited for: its copies may still read the staging blocks)
  if (idx->stage_bytes < want) {
    if (idx->copy_stream) FFX_HIP(hipStreamSynchronize(idx->copy_stream));  // nothing may be in flight out of a block that is released
    for (int i = 0; i < 2; ++i) {
      idx->stage_busy[i] = false;
      if (idx->stage[i]) (void)hipHostFree(idx->stage[i]);
      idx->stage[i] = nullptr;
    }
    idx->stage_bytes = 0;
    for (int i = 0; i < 2; ++i) {
      hipError_t e = hipHostMalloc(&idx->stage[i], want, hipHostMallocDefault);
      if (e != hipSuccess) {
        idx->stage[i] = nullptr;
        return fail(FFX_OOM, "hipHostMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
      }
    }
    idx->stage_bytes = want;
  }
  if (!idx->copy_stream) FFX_HIP(hipStreamCreateWithFlags(&idx->copy_stream, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i)
    if (!idx->stage_ev[i]) FFX_HIP(hipEventCreateWithFlags(&idx->stage_ev[i], hipEventDisableTiming));
  const int64_t per = (int64_t)(idx->stage_bytes / rb);
  const int T = std::max(1, std::min<int>(n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency() / 2, 64));
  // Helper threads live for the whole call (starting a set of threads per 64 MiB block cost about as much as copying the
  // block): block k is gathered by all of them, slice by slice, once the main thread has seen the copy out of its
  // staging block finish (go >= k); the main thread takes slice 0, queues the block's copy when every slice is in
  // (done[k]) and moves on, so the gather of block k + 1 overlaps the copy of block k.
  const int64_t n_blocks = (n_rows + per - 1) / per;
  const int t_use = (int)std::min<int64_t>(T, std::max<int64_t>(1, (int64_t)(std::min(per, n_rows) * (int64_t)rb) >> 20));  // >= 1 MiB per thread and block
  std::atomic<int64_t> go(-1);
  std::atomic<bool> give_up(false);
  std::unique_ptr<std::atomic<int>[]> done(new (std::nothrow) std::atomic<int>[(size_t)n_blocks]);
  if (!done) return fail(FFX_OOM, "host allocation failed");
  for (int64_t k = 0; k < n_blocks; ++k) done[(size_t)k].store(0);
  auto gather = [&](int64_t k, int t) {
    const int64_t b0 = k * per, nb = std::min(per, n_rows - b0);
    char* dst = static_cast<char*>(idx->stage[k & 1]);
    for (int64_t i = nb * t / t_use; i < nb * (t + 1) / t_use; ++i) {
      const int64_t r = rows[b0 + i];
      memcpy(dst + (size_t)i * rb, static_cast<const char*>(chunks[r / chunk_rows]) + (size_t)(r % chunk_rows) * rb, rb);
    }
  };
  auto helper = [&](int t) {
    for (int64_t k = 0; k < n_blocks; ++k) {
      for (unsigned spin = 0; go.load(std::memory_order_acquire) < k; ++spin) {
        if (give_up.load()) return;
        // (waiting for a staging block's copy to finish: up to a millisecond -- sleep, a dozen spinning threads
        // would eat the CPU quota the gather itself needs)
        if (spin > 256) std::this_thread::sleep_for(std::chrono::microseconds(20));
      }
      if (give_up.load()) return;
      gather(k, t);
      done[(size_t)k].fetch_add(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> pool;
  pool.reserve((size_t)std::max(0, t_use - 1));
  try {
    for (int t = 1; t < t_use; ++t) pool.emplace_back(helper, t);
  } catch (const std::system_error&) {
    // fewer helpers than hoped for: the main thread takes the slices nobody was started for
  }
  const int helpers = (int)pool.size();
  int rc = FFX_OK;
  for (int64_t k = 0; k < n_blocks && rc == FFX_OK; ++k) {
    const int s = (int)(k & 1);
    if (idx->stage_busy[s]) {
      if (hipEventSynchronize(idx->stage_ev[s]) != hipSuccess) rc = fail(FFX_HIP_ERROR, "waiting for a staging block failed");
      idx->stage_busy[s] = false;
      if (rc != FFX_OK) break;
    }
    go.store(k, std::memory_order_release);
    gather(k, 0);
    for (int t = helpers + 1; t < t_use; ++t) gather(k, t);
    for (unsigned spin = 0; done[(size_t)k].load(std::memory_order_acquire) < helpers; ++spin)
      if (spin > 256) std::this_thread::sleep_for(std::chrono::microseconds(5));
    const int64_t b0 = k * per, nb = std::min(per, n_rows - b0);
    if (hipMemcpyAsync(idx->row_ptr(dst_row + b0), idx->stage[s], (size_t)nb * rb, hipMemcpyHostToDevice, idx->copy_stream) != hipSuccess ||
        hipEventRecord(idx->stage_ev[s], idx->copy_stream) != hipSuccess)
      rc = fail(FFX_HIP_ERROR, "copy of a staged block to the device failed: %s", hipGetErrorString(hipGetLastError()));
    else
      idx->stage_busy[s] = true;
  }
  give_up.store(true);
  go.store(n_blocks, std::memory_order_release);
  for (auto& th : pool) th.join();
  if (rc != FFX_OK) return rc;
  idx->pending_row = dst_row;
  idx->pending_rows = n_rows;
  return FFX_OK;
}

// The rows of ffx_index_upload_rows_async pulled by the device: every chunk map must be page-locked (ffx_host_pin), the
// kernel reads the rows out of host memory over PCIe on the copy stream.  No staging block, no host threads.
int ffx_index_pull_rows_async(ffx_index* idx, const void* const* chunks, int64_t n_chunks, int64_t chunk_rows,
                              const int64_t* rows, int64_t n_rows, int64_t dst_row) {
  if (!idx || n_rows < 0 || dst_row < 0 || n_chunks < 0 || chunk_rows <= 0) return fail(FFX_BAD_ARG, "bad argument");
  if (n_rows == 0) return FFX_OK;
  if (!chunks || !rows) return fail(FFX_BAD_ARG, "NULL argument");
  if (dst_row + n_rows > idx->capacity) return fail(FFX_BAD_ARG, "rows [%lld, %lld) exceed the capacity %lld", (long long)dst_row, (long long)(dst_row + n_rows), (long long)idx->capacity);
  const int64_t n_src = n_chunks * chunk_rows;
  for (int64_t i = 0; i < n_rows; ++i)
    if (rows[i] < 0 || rows[i] >= n_src) return fail(FFX_BAD_ARG, "rows[%lld] = %lld is outside the %lld stored rows", (long long)i, (long long)rows[i], (long long)n_src);
  FFX_HIP(hipSetDevice(idx->device));
  if (!idx->copy_stream) FFX_HIP(hipStreamCreateWithFlags(&idx->copy_stream, hipStreamNonBlocking));
  if (idx->pending_rows > 0) FFX_TRY(ffx_index_upload_wait(idx));  // (an earlier asynchronous upload nobody waited for)
  const size_t rb = (size_t)idx->dim * idx->elem;
  size_t align = 16;
  while (align > 1 && rb % align) align >>= 1;
  if (idx->pull_chunk_hosts.size() != (size_t)n_chunks || !std::equal(chunks, chunks + n_chunks, idx->pull_chunk_hosts.begin())) {
    std::vector<const void*> dev((size_t)n_chunks);
    for (int64_t c = 0; c < n_chunks; ++c) {
      void* d = nullptr;
      if (!chunks[c] || hipHostGetDevicePointer(&d, const_cast<void*>(chunks[c]), 0) != hipSuccess || !d) {
        (void)hipGetLastError();
        return fail(FFX_BAD_ARG, "chunk %lld is not page-locked host memory (ffx_host_pin it first)", (long long)c);
      }
      dev[(size_t)c] = d;
    }
    idx->pull_chunk_hosts.clear();
    FFX_TRY(idx->pull_chunks.ensure((size_t)n_chunks * sizeof(void*)));
    FFX_HIP(hipStreamSynchronize(idx->copy_stream));  // (a kernel of an earlier call may still read the old table)
    FFX_TRY(ffx_h2d_now(idx->device, idx->pull_chunks.p, dev.data(), (size_t)n_chunks * sizeof(void*)));
    idx->pull_chunk_hosts.assign(chunks, chunks + n_chunks);
  }
  for (int64_t c = 0; c < n_chunks; ++c)  // the widest unit every row starts on
    while (align > 1 && reinterpret_cast<uintptr_t>(chunks[c]) % align) align >>= 1;
  FFX_TRY(idx->pull_list.ensure((size_t)n_rows * sizeof(int64_t)));
  if (idx->pull_list_room < n_rows) {  // (nothing is in flight out of the old block: pending_rows was 0 or has been waited for)
    if (idx->pull_list_host) (void)hipHostFree(idx->pull_list_host);
    idx->pull_list_host = nullptr;
    idx->pull_list_room = 0;
    const int64_t room = std::max<int64_t>(n_rows + n_rows / 4, 1 << 16);
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&idx->pull_list_host), (size_t)room * sizeof(int64_t), hipHostMallocDefault);
    if (e != hipSuccess) {
      idx->pull_list_host = nullptr;
      return fail(FFX_OOM, "hipHostMalloc(%zu bytes) failed: %s", (size_t)room * sizeof(int64_t), hipGetErrorString(e));
    }
    idx->pull_list_room = room;
  }
  memcpy(idx->pull_list_host, rows, (size_t)n_rows * sizeof(int64_t));
  FFX_HIP(hipMemcpyAsync(idx->pull_list.p, idx->pull_list_host, (size_t)n_rows * sizeof(int64_t), hipMemcpyHostToDevice, idx->copy_stream));
  const int units = (int)(rb / align);
  int G = 64;
  while (G > 1 && G / 2 >= units) G >>= 1;
  const int64_t waves = (n_rows + 64 / G - 1) / (64 / G);
  // 64 workgroups keep the link full (windows of a streamed index: 54 GB/s; 32: 51, 16: 35) and hardly slow the scoring
  // kernels running beside them (0.75 vs 0.55 ms); 256 move no more bytes and slow those kernels 10 - 15 x -- their
  // outstanding reads fill the queues every kernel's loads go through (tools/probes/host_pull_probe.hip: an HBM-bound
  // kernel beside a pull of 16 / 64 / 256 workgroups runs 1.6 / 5.5 / 8 x longer)
  int64_t grid_cap = 64;
  if (long long v = 0; ffx_experiment("pull_grid", &v)) grid_cap = std::max<long long>(1, v);
  const int grid = (int)std::max<int64_t>(1, std::min<int64_t>(grid_cap, (waves + 3) / 4));
  const char* const* table = idx->pull_chunks.as<const char*>();
  char* dst = static_cast<char*>(idx->row_ptr(dst_row));
  if (ffx_trace_host()) fprintf(stderr, "\nffx trace: pull %lld rows, chunks[0]=%p (%lld x %lld rows of %zu bytes), list_host=%p list_dev=%p table_dev=%p dst=%p..%p\n", (long long)n_rows, chunks[0], (long long)n_chunks, (long long)chunk_rows, rb, (void*)idx->pull_list_host, idx->pull_list.p, (const void*)table, (void*)dst, (void*)(dst + (size_t)n_rows * rb));
  if (align == 16)
    ffx::pull_rows_kernel<uint4><<<grid, 256, 0, idx->copy_stream>>>(table, chunk_rows, idx->pull_list.as<int64_t>(), n_rows, units, G, dst);
  else if (align == 8)  // (where an HDF5 file's allocator usually puts a chunk)
    ffx::pull_rows_kernel<uint2><<<grid, 256, 0, idx->copy_stream>>>(table, chunk_rows, idx->pull_list.as<int64_t>(), n_rows, (int)(rb / 8), G = 64, dst);
  else if (align >= 4)
    ffx::pull_rows_kernel<uint32_t><<<grid, 256, 0, idx->copy_stream>>>(table, chunk_rows, idx->pull_list.as<int64_t>(), n_rows, (int)(rb / 4), G = 64, dst);
  else
    ffx::pull_rows_kernel<unsigned char><<<grid, 256, 0, idx->copy_stream>>>(table, chunk_rows, idx->pull_list.as<int64_t>(), n_rows, (int)rb, G = 64, dst);
  FFX_HIP(hipGetLastError());
  idx->pending_row = dst_row;
  idx->pending_rows = n_rows;
  return FFX_OK;
}

// Orders what is queued on idx's copy stream from now on behind what `before`'s copy stream holds so far: the windows
// of a streamed index are filled one after the other instead of sharing the link (the first one is ready -- and
// scored, and free again -- while the second is still on its way).
int ffx_index_upload_after(ffx_index* idx, ffx_index* before) {
  if (!idx || !before) return fail(FFX_BAD_ARG, "NULL argument");
  if (idx == before || !before->copy_stream) return FFX_OK;
  if (idx->device != before->device) return fail(FFX_BAD_ARG, "the indexes live on different devices");
  FFX_HIP(hipSetDevice(idx->device));
  if (!idx->copy_stream) FFX_HIP(hipStreamCreateWithFlags(&idx->copy_stream, hipStreamNonBlocking));
  if (!before->order_ev) FFX_HIP(hipEventCreateWithFlags(&before->order_ev, hipEventDisableTiming));
  FFX_HIP(hipEventRecord(before->order_ev, before->copy_stream));
  FFX_HIP(hipStreamWaitEvent(idx->copy_stream, before->order_ev, 0));
  return FFX_OK;
}

int ffx_index_upload_wait(ffx_index* idx) {
  if (!idx) return fail(FFX_BAD_ARG, "NULL argument");
  if (!idx->copy_stream) return FFX_OK;
  FFX_HIP(hipSetDevice(idx->device));
  FFX_HIP(hipStreamSynchronize(idx->copy_stream));
  idx->stage_busy[0] = idx->stage_busy[1] = false;
  const int64_t row = idx->pending_row, n = idx->pending_rows;
  idx->pending_rows = 0;
  return n > 0 ? check_pq_codes(idx, row, n) : FFX_OK;
}

int ffx_index_copy_rows(ffx_index* dst, const ffx_index* src, int64_t n_rows) {
  if (!dst || !src) return fail(FFX_BAD_ARG, "NULL argument");
  if (dst->device != src->device || dst->dim != src->dim || dst->dtype != src->dtype)
    return fail(FFX_BAD_ARG, "indexes differ in device, dim or dtype");
  if (n_rows < 0 || n_rows > src->capacity || n_rows > dst->capacity) return fail(FFX_BAD_ARG, "n_rows out of range");
  if (n_rows == 0) return FFX_OK;
  FFX_HIP(hipSetDevice(dst->device));
  FFX_HIP(hipMemcpy(dst->vectors, src->vectors, (size_t)n_rows * src->dim * src->elem, hipMemcpyDeviceToDevice));
  // a device-to-device hipMemcpy may return before the copy is done, and the plans' streams are
  // non-blocking (not ordered after the null stream): wait here, so callers may use either index freely
  FFX_HIP(hipDeviceSynchronize());
  return FFX_OK;
}

// Rows back to the host, THROUGH A PAGE-LOCKED BLOCK OF THE INDEX'S OWN (8 MiB, a chunk of rows at a time) and a host memcpy.
// Handing the caller's ordinary memory to hipMemcpy makes the runtime page-lock those pages on the fly for transfers of more
// than 1 MiB, and that is where the one abort this library has seen comes from: in 6 of 21 full runs of the GPU suite of round
// 4 the runtime ended the process from its own thread with "Memory access fault by GPU node-N on address A. Reason: Write
// access to a read-only page" -- every time with A inside the 1.32 MB NumPy array (heap memory, part of it never written before)
// that this function was copying into at that moment (FFX_TRACE_HOST=1; profiles/r04/abort_hunt).  Nothing of the sort when
// the device writes into page-locked memory and the CPU writes the caller's.
int ffx_index_gather(ffx_index* idx, const int32_t* rows, int64_t n_rows, void* host_out) {
  if (!idx || (n_rows > 0 && (!rows || !host_out))) return fail(FFX_BAD_ARG, "NULL argument");
  if (n_rows == 0) return FFX_OK;
  for (int64_t i = 0; i < n_rows; ++i)
    if (rows[i] < 0 || rows[i] >= idx->capacity) return fail(FFX_BAD_ARG, "row %d outside the index", rows[i]);
  FFX_HIP(hipSetDevice(idx->device));
  const size_t row_bytes = (size_t)idx->dim * idx->elem;
  const int64_t per_chunk = std::max<int64_t>(1, std::min<int64_t>(n_rows, (int64_t)((size_t(8) << 20) / std::max<size_t>(row_bytes, 1))));
  const size_t list_bytes = ((size_t)per_chunk * sizeof(int32_t) + 255) & ~size_t(255);
  const size_t stage_need = list_bytes + (size_t)per_chunk * row_bytes;
  if (idx->gather_stage_bytes < stage_need) {
    if (idx->gather_stage) (void)hipHostFree(idx->gather_stage);
    idx->gather_stage = nullptr;
    idx->gather_stage_bytes = 0;
    const hipError_t e = hipHostMalloc(&idx->gather_stage, stage_need, hipHostMallocDefault);
    if (e != hipSuccess) {
      idx->gather_stage = nullptr;
      (void)hipGetLastError();
      return fail(FFX_OOM, "hipHostMalloc(%zu bytes) failed: %s", stage_need, hipGetErrorString(e));
    }
    idx->gather_stage_bytes = stage_need;
  }
  char* s_rows = static_cast<char*>(idx->gather_stage);
  char* s_out = s_rows + list_bytes;
  if (ffx_trace_host()) fprintf(stderr, "\nffx trace: gather %lld rows of %zu bytes -> %p through the page-locked block %p\n", (long long)n_rows, row_bytes, host_out, idx->gather_stage);
  DevBuf drows, dout;
  int rc = drows.ensure((size_t)per_chunk * sizeof(int32_t));
  if (rc == FFX_OK) rc = dout.ensure((size_t)per_chunk * row_bytes);
  for (int64_t c0 = 0; c0 < n_rows && rc == FFX_OK; c0 += per_chunk) {
    const int64_t m = std::min(per_chunk, n_rows - c0);
    memcpy(s_rows, rows + c0, (size_t)m * sizeof(int32_t));
    hipError_t e = hipMemcpy(drows.p, s_rows, (size_t)m * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      // rows are copied as raw bytes, so one kernel serves every storage type
      if (row_bytes % 2 == 0)
        ffx::gather_rows_kernel<uint16_t><<<blocks_for(m, 4), 256>>>(static_cast<const uint16_t*>(idx->vectors), (int)(row_bytes / 2), drows.as<int32_t>(), m,
                                                                     dout.as<uint16_t>());
      else
        ffx::gather_rows_kernel<uint8_t><<<blocks_for(m, 4), 256>>>(static_cast<const uint8_t*>(idx->vectors), (int)row_bytes, drows.as<int32_t>(), m, dout.as<uint8_t>());
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(s_out, dout.p, (size_t)m * row_bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(FFX_HIP_ERROR, "gather failed: %s", hipGetErrorString(e));
    if (rc == FFX_OK) memcpy(static_cast<char*>(host_out) + (size_t)c0 * row_bytes, s_out, (size_t)m * row_bytes);
  }
  drows.release();
  dout.release();
  return rc;
}

int ffx_index_fill_synthetic(ffx_index* idx, uint64_t seed, float scale, int64_t row_offset, int64_t n_rows,
                             int64_t first_global_row) {
  if (!idx) return fail(FFX_BAD_ARG, "NULL argument");
  if (row_offset < 0 || n_rows < 0 || row_offset + n_rows > idx->capacity || first_global_row < 0)
    return fail(FFX_BAD_ARG, "rows outside capacity");
  if (n_rows == 0) return FFX_OK;
  FFX_HIP(hipSetDevice(idx->device));
  const int64_t n_elems = n_rows * idx->dim;
  const unsigned grid = (unsigned)std::min<int64_t>((n_elems + 255) / 256, 256 * 32);
  if (idx->dtype == FFX_DTYPE_PQ8)
    ffx::fill_codes_kernel<<<grid, 256>>>(reinterpret_cast<uint8_t*>(idx->row_ptr(row_offset)), n_elems,
                                          (uint64_t)first_global_row * (uint64_t)idx->dim, seed, idx->pq_ks > 0 ? idx->pq_ks : 256);
  else if (idx->dtype == FFX_DTYPE_F16)
    ffx::fill_synthetic_kernel<_Float16><<<grid, 256>>>(reinterpret_cast<_Float16*>(idx->row_ptr(row_offset)), n_elems,
                                                        (uint64_t)first_global_row * (uint64_t)idx->dim, seed, scale);
  else
    ffx::fill_synthetic_kernel<float><<<grid, 256>>>(reinterpret_cast<float*>(idx->row_ptr(row_offset)), n_elems,
                                                     (uint64_t)first_global_row * (uint64_t)idx->dim, seed, scale);
  FFX_HIP(hipGetLastError());
  FFX_HIP(hipDeviceSynchronize());
  return FFX_OK;
}

int ffx_index_set_doc_rows(ffx_index* idx, const int64_t* doc_off, int64_t n_docs, const int32_t* doc_rows,
                           int64_t n_doc_rows) {
  if (!idx || n_docs < 0 || n_doc_rows < 0 || (n_docs > 0 && !doc_off) || (n_doc_rows > 0 && !doc_rows))
    return fail(FFX_BAD_ARG, "bad argument");
  if (n_docs > INT32_MAX - 1 || n_doc_rows > INT32_MAX) return fail(FFX_BAD_ARG, "too many documents/rows for int32 ids");
  std::vector<int32_t> off32((size_t)n_docs + 1, 0), first((size_t)std::max<int64_t>(n_docs, 1), 0);
  bool identity = n_docs == n_doc_rows;
  if (n_docs > 0) {
    if (doc_off[0] != 0 || doc_off[n_docs] != n_doc_rows) return fail(FFX_BAD_ARG, "doc_off must run from 0 to n_doc_rows");
    for (int64_t i = 0; i < n_docs; ++i) {
      const int64_t c = doc_off[i + 1] - doc_off[i];
      if (c < 0) return fail(FFX_BAD_ARG, "doc_off not monotone at %lld", (long long)i);
      off32[(size_t)i + 1] = (int32_t)doc_off[i + 1];
    }
  }
  for (int64_t i = 0; i < n_doc_rows; ++i)
    if (doc_rows[i] < 0 || doc_rows[i] >= idx->capacity) return fail(FFX_BAD_ARG, "doc_rows[%lld] outside the index", (long long)i);
  for (int64_t i = 0; i < n_docs; ++i) {
    const bool has = doc_off[i + 1] > doc_off[i];
    first[(size_t)i] = has ? doc_rows[doc_off[i]] : -1;
    if (!has || doc_off[i] != i || doc_rows[i] != i) identity = false;
  }
  FFX_HIP(hipSetDevice(idx->device));
  FFX_TRY(idx->doc_off.ensure(off32.size() * sizeof(int32_t)));
  FFX_TRY(idx->doc_rows.ensure(std::max<size_t>((size_t)n_doc_rows, 1) * sizeof(int32_t)));
  FFX_TRY(idx->doc_first.ensure(first.size() * sizeof(int32_t)));
  FFX_TRY(ffx_h2d_now(idx->device, idx->doc_off.p, off32.data(), off32.size() * sizeof(int32_t)));
  FFX_TRY(ffx_h2d_now(idx->device, idx->doc_first.p, first.data(), first.size() * sizeof(int32_t)));
  if (n_doc_rows > 0)
    FFX_TRY(ffx_h2d_now(idx->device, idx->doc_rows.p, doc_rows, (size_t)n_doc_rows * sizeof(int32_t)));
  idx->n_docs = n_docs;
  idx->n_doc_rows = n_doc_rows;
  idx->max_doc_rows = 1;
  for (int64_t i = 0; i < n_docs; ++i) idx->max_doc_rows = std::max(idx->max_doc_rows, doc_off[i + 1] - doc_off[i]);
  idx->docs_identity = identity && n_docs > 0;
  idx->doc_rows_identity = n_doc_rows > 0;
  for (int64_t i = 0; i < n_doc_rows && idx->doc_rows_identity; ++i) idx->doc_rows_identity = doc_rows[i] == i;
  idx->docs_nonempty = n_docs > 0;
  for (int64_t i = 0; i < n_docs && idx->docs_nonempty; ++i) idx->docs_nonempty = doc_off[i + 1] > doc_off[i];
  return FFX_OK;
}

int ffx_index_set_psg_rows(ffx_index* idx, const int32_t* psg_row, int64_t n_psgs) {
  if (!idx || n_psgs < 0 || (n_psgs > 0 && !psg_row)) return fail(FFX_BAD_ARG, "bad argument");
  if (n_psgs > INT32_MAX) return fail(FFX_BAD_ARG, "too many passages for int32 ids");
  FFX_HIP(hipSetDevice(idx->device));
  FFX_TRY(idx->psg_row.ensure(std::max<size_t>((size_t)n_psgs, 1) * sizeof(int32_t)));
  if (n_psgs > 0) FFX_TRY(ffx_h2d_now(idx->device, idx->psg_row.p, psg_row, (size_t)n_psgs * sizeof(int32_t)));
  idx->n_psgs = n_psgs;
  bool identity = n_psgs > 0 && n_psgs <= idx->capacity;
  for (int64_t i = 0; i < n_psgs && identity; ++i) identity = psg_row[i] == i;
  idx->psgs_identity = identity;
  return FFX_OK;
}

// ---- plan -------------------------------------------------------------------------------------
// with_stream = false: a plan that only ever runs on streams lent to it (slices 1 and 2 of ffx_score)
static int plan_create(ffx_index* idx, ffx_plan** out, bool with_stream) {
  if (!idx || !out) return fail(FFX_BAD_ARG, "NULL argument");
  *out = nullptr;
  FFX_HIP(hipSetDevice(idx->device));
  ffx_plan* pl = new (std::nothrow) ffx_plan();
  if (!pl) return fail(FFX_OOM, "host allocation failed");
  pl->idx = idx;
  pl->device = idx->device;
  hipError_t e = with_stream ? hipStreamCreateWithFlags(&pl->own_stream, hipStreamNonBlocking) : hipSuccess;
  pl->stream = pl->own_stream;
  // The stage-timing events (0 .. 3, and the ring of gather_dot brackets) only ever feed hipEventElapsedTime: created WITHOUT
  // the system-scope fence a default event performs when it is recorded (cache write-back and invalidation, and what
  // those cost the kernel behind it: ~0.08 ms per record, 0.55 ms of a 17.3 ms step with seven records between the kernels).
  // Event 4 closes a run: other streams and the host wait on it before results are read, so it keeps the fence.
  for (int i = 0; i < 5 && e == hipSuccess; ++i) e = i < 4 ? hipEventCreateWithFlags(&pl->ev[i], hipEventDisableSystemFence) : hipEventCreate(&pl->ev[i]);
  if (e != hipSuccess) {
    ffx_plan_destroy(pl);
    return fail(FFX_HIP_ERROR, "stream/event creation failed: %s", hipGetErrorString(e));
  }
  if (idx->qdim <= 0) {
    ffx_plan_destroy(pl);
    return fail(FFX_BAD_ARG, "set the codebook (ffx_index_set_codebook) before creating a plan");
  }
  // leaf plan of the exact generic kernel depends on the (decoded) dimensionality only
  std::vector<int> bounds;
  numpy_leaf_plan(0, idx->qdim, bounds);
  bounds.push_back(idx->qdim);
  pl->n_leaves = (int)bounds.size() - 1;
  pl->anyd_ok = true;
  for (size_t l = 0; l + 1 < bounds.size(); ++l) {
    const int len = bounds[l + 1] - bounds[l];
    pl->anyd_ok = pl->anyd_ok && len >= 8 && len <= 128;
    pl->anyd_tails = pl->anyd_tails || (len % 8) != 0;
  }
  int rc = pl->leaf_start.ensure(bounds.size() * sizeof(int));
  if (rc == FFX_OK && ffx_h2d_now(pl->device, pl->leaf_start.p, bounds.data(), bounds.size() * sizeof(int)) != FFX_OK)
    rc = fail(FFX_HIP_ERROR, "leaf plan upload failed");
  if (rc == FFX_OK) {
    std::vector<int> steps;
    int next_leaf = 0;
    numpy_combine_plan(idx->qdim, next_leaf, steps);
    steps.push_back(0);  // never empty
    rc = pl->combine_steps.ensure(steps.size() * sizeof(int));
    if (rc == FFX_OK && ffx_h2d_now(pl->device, pl->combine_steps.p, steps.data(), steps.size() * sizeof(int)) != FFX_OK)
      rc = fail(FFX_HIP_ERROR, "combine plan upload failed");
  }
  // (experiments build, FFX_DEBUG_VARIANT=seg_leaves=N: ANY row is cut into pieces of at most N leaves of NumPy's tree and scored
  // piece by piece -- d = 768 with N = 1: eight passes over 96-element slices, every pass's query slices 2.7 MB in all)
  long long seg_leaves = 0;
  const bool forced_pieces = ffx_experiment("seg_leaves", &seg_leaves) && seg_leaves >= 1 && seg_leaves < 8 && pl->n_leaves > seg_leaves;
  if (rc == FFX_OK && (pl->n_leaves > 8 || forced_pieces) && idx->dtype != FFX_DTYPE_PQ8) {
    // more than 8 leaves: segments of <= 8 leaves, each with its own leaf table and combine steps
    std::vector<std::pair<int, int>> pieces;
    std::vector<int> above, tables;
    g_max_piece_leaves = forced_pieces ? (int)seg_leaves : 8;
    long_row_plan(0, idx->qdim, pieces, above);
    g_max_piece_leaves = 8;
    pl->forced_pieces = forced_pieces;
    bool ok = pieces.size() <= 64;
    for (const auto& [col0, len] : pieces) {
      ffx_plan::Segment sg{};
      sg.col0 = col0;
      sg.len = len;
      std::vector<int> b;
      numpy_leaf_plan(0, len, b);
      b.push_back(len);
      sg.n_leaves = (int)b.size() - 1;
      for (size_t l = 0; l + 1 < b.size(); ++l) {
        const int ll = b[l + 1] - b[l];
        ok = ok && ll >= 8 && ll <= 128;
        sg.tails = sg.tails || (ll % 8) != 0;
      }
      sg.table_at = (int)tables.size();
      tables.insert(tables.end(), b.begin(), b.end());
      std::vector<int> st;
      int next = 0;
      numpy_combine_plan(len, next, st);
      st.push_back(0);
      sg.steps_at = (int)tables.size();
      tables.insert(tables.end(), st.begin(), st.end());
      pl->segments.push_back(sg);
    }
    pl->seg_steps_at = (int)tables.size();
    pl->seg_n_steps = (int)above.size() / 2;
    tables.insert(tables.end(), above.begin(), above.end());
    tables.push_back(0);
    if (!ok) {
      pl->segments.clear();  // (rows beyond ~64 K elements: the staged generic kernel)
    } else {
      rc = pl->seg_tables.ensure(tables.size() * sizeof(int));
      if (rc == FFX_OK && ffx_h2d_now(pl->device, pl->seg_tables.p, tables.data(), tables.size() * sizeof(int)) != FFX_OK)
        rc = fail(FFX_HIP_ERROR, "segment plan upload failed");
    }
  }
  if (rc == FFX_OK) rc = pl->flags.ensure(64);
  if (rc != FFX_OK) {
    ffx_plan_destroy(pl);
    return rc;
  }
  *out = pl;
  return FFX_OK;
}

int ffx_plan_create(ffx_index* idx, ffx_plan** out) { return plan_create(idx, out, true); }

int ffx_plan_destroy(ffx_plan* pl) {
  if (!pl) return FFX_OK;
  (void)hipSetDevice(pl->device);
  if (pl->stream) (void)hipStreamSynchronize(pl->stream);
  if (pl->es_child) {
    pl->es_child->stream = pl->es_child->own_stream;  // it borrowed this plan's stream
    ffx_plan_destroy(pl->es_child);
    pl->es_child = nullptr;
  }
  for (DevBuf* b : {&pl->es_active, &pl->es_cnt, &pl->es_pos, &pl->es_scan_tmp, &pl->small_flag}) b->release();
  for (DevBuf* b : {&pl->queries, &pl->pair_off, &pl->pair_ids, &pl->sparse, &pl->cnt, &pl->seg_off, &pl->desc,
                    &pl->rs, &pl->scan_tmp, &pl->sort_tmp, &pl->iota, &pl->sorted_keys, &pl->leaf_start, &pl->lut, &pl->vdesc, &pl->rs_v, &pl->combine_steps, &pl->seg_tables,
                    &pl->flags, &pl->ff, &pl->interp, &pl->order, &pl->carry, &pl->ranked, &pl->carry_ranked})
    b->release();
  for (auto& e : pl->ev)
    if (e) (void)hipEventDestroy(e);
  for (auto& pair : pl->recent)
    for (auto& e : pair)
      if (e) (void)hipEventDestroy(e);
  if (pl->own_stream) (void)hipStreamDestroy(pl->own_stream);
  for (int l = 0; l < ffx_plan::kOutLanes; ++l) {
    for (int i = 0; i < 2; ++i) {
      if (pl->out_stage[l][i]) (void)hipHostFree(pl->out_stage[l][i]);
      if (pl->out_stage_ev[l][i]) (void)hipEventDestroy(pl->out_stage_ev[l][i]);
    }
  }
  if (pl->hstage) (void)hipHostFree(pl->hstage);
  delete pl;
  return FFX_OK;
}

int ffx_plan_set_stream(ffx_plan* pl, void* stream) {
  if (!pl) return fail(FFX_BAD_ARG, "NULL plan");
  FFX_HIP(hipSetDevice(pl->idx->device));
  FFX_HIP(hipStreamSynchronize(pl->stream));
  pl->stream = stream ? (hipStream_t)stream : pl->own_stream;
  return FFX_OK;
}

int ffx_plan_set_queries(ffx_plan* pl, const float* host_queries, int64_t n_queries) {
  if (!pl || n_queries < 0 || (n_queries > 0 && !host_queries)) return fail(FFX_BAD_ARG, "bad argument");
  if (n_queries > INT32_MAX) return fail(FFX_BAD_ARG, "too many queries");
  FFX_HIP(hipSetDevice(pl->idx->device));
  const size_t bytes = (size_t)n_queries * pl->idx->qdim * sizeof(float);
  FFX_TRY(pl->queries.ensure(std::max<size_t>(bytes, 16)));
  pl->hstage_used = 0;
  FFX_TRY(plan_h2d(pl, pl->queries.p, host_queries, bytes));
  FFX_HIP(hipStreamSynchronize(pl->stream));
  pl->n_queries_set = (int)n_queries;
  pl->ran = false;
  return FFX_OK;
}

int ffx_plan_set_queries_device(ffx_plan* pl, const float* dev_queries, int64_t n_queries) {
  if (!pl || n_queries < 0 || (n_queries > 0 && !dev_queries)) return fail(FFX_BAD_ARG, "bad argument");
  if (n_queries > INT32_MAX) return fail(FFX_BAD_ARG, "too many queries");
  FFX_HIP(hipSetDevice(pl->device));
  const size_t bytes = (size_t)n_queries * pl->idx->qdim * sizeof(float);
  FFX_TRY(pl->queries.ensure(std::max<size_t>(bytes, 16)));
  if (bytes) FFX_HIP(hipMemcpyAsync(pl->queries.p, dev_queries, bytes, hipMemcpyDeviceToDevice, pl->stream));
  FFX_HIP(hipStreamSynchronize(pl->stream));
  pl->n_queries_set = (int)n_queries;
  pl->ran = false;
  return FFX_OK;
}

static int launch_count_scan(ffx_plan* pl) {
  const ffx_index* ix = pl->idx;
  const int64_t P = pl->n_pairs;
  ffx::count_rows_kernel<<<blocks_for(P + 1, 256), 256, 0, pl->stream>>>(
      pl->pair_ids.as<int32_t>(), P, pl->mode, ix->doc_off.as<int32_t>(), ix->n_docs, ix->psg_row.as<int32_t>(),
      ix->n_psgs, ix->capacity, pl->cnt.as<int32_t>(), pl->flags.as<unsigned long long>());
  FFX_HIP(hipGetLastError());
  size_t tmp = 0;
  FFX_HIP(rocprim::exclusive_scan(nullptr, tmp, pl->cnt.as<int32_t>(), pl->seg_off.as<int32_t>(), (int32_t)0,
                                  (size_t)(P + 1), rocprim::plus<int32_t>(), pl->stream));
  FFX_TRY(pl->scan_tmp.ensure(std::max<size_t>(tmp, 16)));
  FFX_HIP(rocprim::exclusive_scan(pl->scan_tmp.p, tmp, pl->cnt.as<int32_t>(), pl->seg_off.as<int32_t>(), (int32_t)0,
                                  (size_t)(P + 1), rocprim::plus<int32_t>(), pl->stream));
  return FFX_OK;
}

static int finish_set_pairs(ffx_plan* pl, int64_t* bad_pair);

int ffx_plan_set_pairs(ffx_plan* pl, const int64_t* pair_off, int64_t n_queries, const int32_t* pair_ids,
                       const float* sparse, int mode, int64_t* bad_pair) {
  if (bad_pair) *bad_pair = -1;
  if (!pl || !pair_off || n_queries < 0) return fail(FFX_BAD_ARG, "bad argument");
  if (mode < FFX_MODE_PASSAGE || mode > FFX_MODE_AVEP) return fail(FFX_BAD_ARG, "unknown mode %d", mode);
  if (n_queries > INT32_MAX - 1) return fail(FFX_BAD_ARG, "too many queries");
  if (pair_off[0] != 0) return fail(FFX_BAD_ARG, "pair_off[0] must be 0");
  for (int64_t q = 0; q < n_queries; ++q)
    if (pair_off[q + 1] < pair_off[q]) return fail(FFX_BAD_ARG, "pair_off not monotone at query %lld", (long long)q);
  const int64_t P = pair_off[n_queries];
  if (P > INT32_MAX - 1) return fail(FFX_BAD_ARG, "more than 2^31-2 pairs in one batch: split the queries");
  if (P > 0 && !pair_ids) return fail(FFX_BAD_ARG, "pair_ids is NULL");
  ffx_index* ix = pl->idx;
  if (mode == FFX_MODE_PASSAGE ? ix->n_psgs == 0 : ix->n_docs == 0) {
    if (P > 0) {
      if (bad_pair) *bad_pair = 0;
      return fail(FFX_NOT_FOUND, "the index holds no %s ids", mode == FFX_MODE_PASSAGE ? "passage" : "document");
    }
  }
  FFX_HIP(hipSetDevice(ix->device));
  pl->have_pairs = false;
  pl->dense_given = false;
  pl->ran = false;
  pl->n_queries = n_queries;
  pl->n_pairs = P;
  pl->mode = mode;
  pl->have_sparse = sparse != nullptr;
  pl->host_pair_off.assign(pair_off, pair_off + n_queries + 1);
  pl->q_hint_valid = false;  // (the hint goes with the offsets it was made from: upload_query_hint makes a new one)
  const size_t pb = std::max<size_t>((size_t)P, 1);
  FFX_TRY(pl->pair_off.ensure((size_t)(n_queries + 1) * sizeof(int64_t)));
  FFX_TRY(pl->pair_ids.ensure(pb * sizeof(int32_t)));
  FFX_TRY(pl->cnt.ensure((pb + 1) * sizeof(int32_t)));
  FFX_TRY(pl->seg_off.ensure((pb + 1) * sizeof(int32_t)));
  FFX_TRY(pl->ff.ensure(pb * sizeof(float)));
  if (sparse) {
    FFX_TRY(pl->sparse.ensure(pb * sizeof(float)));
    FFX_TRY(pl->interp.ensure(pb * sizeof(float)));
  }
  pl->hstage_used = 0;
  FFX_TRY(plan_h2d(pl, pl->pair_off.p, pair_off, (size_t)(n_queries + 1) * sizeof(int64_t)));
  if (P > 0) {
    const size_t pbytes = (size_t)P * sizeof(int32_t);
    if (sparse && pbytes > (1u << 20)) {  // both arrays above the small-copy block: staged together (one round of threads)
      void* dsts[2] = {pl->pair_ids.p, pl->sparse.p};
      const void* srcs[2] = {pair_ids, sparse};
      const size_t sizes[2] = {pbytes, pbytes};
      FFX_TRY(ffx_h2d_many(pl->device, 2, dsts, srcs, sizes, pl->stream));
    } else {
      FFX_TRY(plan_h2d(pl, pl->pair_ids.p, pair_ids, pbytes));
      if (sparse) FFX_TRY(plan_h2d(pl, pl->sparse.p, sparse, pbytes));
    }
  }
  return finish_set_pairs(pl, bad_pair);
}

// the second half of ffx_plan_set_pairs: the lists are in the plan's device buffers; validate the ordinals and
// size the row list on the device (get_indices, index/util.py:29-41).  In two steps, so that a caller with more to queue
// on the stream (the sharded scorer's staging) waits once for everything: launch (asynchronous; the verdict and, where
// candidates may contribute several rows, the row count travel into page-locked memory) and complete (after the stream
// has been synchronised).
static int upload_query_hint(ffx_plan* pl) {
  const int64_t P = pl->n_pairs, Q = pl->n_queries;
  pl->q_hint_valid = false;
  if (P <= 0 || Q <= 0 || !derive_kernel_exists(pl) || (int64_t)pl->host_pair_off.size() != Q + 1) return FFX_OK;
  const int64_t nb = (P + ffx_plan::kHintBlock - 1) / ffx_plan::kHintBlock;
  pl->host_hint.resize((size_t)nb + 1);
  const int64_t* off = pl->host_pair_off.data();
  int64_t q = 0;
  for (int64_t b = 0; b < nb; ++b) {
    const int64_t p = b * ffx_plan::kHintBlock;  // (p < P = off[Q]: the walk ends at a query that has candidates)
    while (off[q + 1] <= p) ++q;
    pl->host_hint[(size_t)b] = (int32_t)q;
  }
  pl->host_hint[(size_t)nb] = (int32_t)(Q - 1);
  FFX_TRY(pl->q_hint.ensure(((size_t)nb + 1) * sizeof(int32_t)));
  FFX_TRY(plan_h2d(pl, pl->q_hint.p, pl->host_hint.data(), ((size_t)nb + 1) * sizeof(int32_t)));
  pl->q_hint_valid = true;
  return FFX_OK;
}

static int set_pairs_check_launch(ffx_plan* pl) {
  ffx_index* ix = pl->idx;
  const int64_t P = pl->n_pairs;
  const int mode = pl->mode;
  FFX_TRY(upload_query_hint(pl));
  FFX_HIP(hipMemsetAsync(pl->flags.p, 0xFF, sizeof(unsigned long long), pl->stream));
  // Where every valid ordinal names exactly one row (passages that are their own rows; FirstP, or documents of one row
  // each, when no document is empty) there is nothing to count or scan: the ordinals are range-checked and the row
  // list has one entry per candidate.  (The count + scan of a 4 M-pair slice took 0.12 ms beside the scoring kernels of
  // the slice before: a tenth of ffx_score's device time went there.)
  pl->check_one_each = (mode == FFX_MODE_PASSAGE ? ix->psgs_identity : ix->docs_nonempty && (mode == FFX_MODE_FIRSTP || ix->max_doc_rows <= 1)) &&
                       !ffx_experiment("count_always", nullptr);
  if (pl->check_one_each) {
    if (P > 0) {
      ffx::check_range_kernel<<<blocks_for(P, 256), 256, 0, pl->stream>>>(pl->pair_ids.as<int32_t>(), P, (int32_t)(mode == FFX_MODE_PASSAGE ? ix->n_psgs : ix->n_docs),
                                                                          pl->flags.as<unsigned long long>());
      FFX_HIP(hipGetLastError());
    }
  } else {
    FFX_TRY(launch_count_scan(pl));
  }
  char* back = pl->stage(16);  // both read-backs land in page-locked memory: one wait, no blocking copies
  pl->check_back = back ? back : pl->check_host;
  memset(pl->check_back, 0xFF, 8);
  memset(pl->check_back + 8, 0, 8);
  FFX_TRY(ffx_d2h(pl->device, pl->check_back, pl->flags.p, 8, pl->stream));
  if (!pl->check_one_each) FFX_TRY(ffx_d2h(pl->device, pl->check_back + 8, pl->seg_off.as<int32_t>() + P, sizeof(int32_t), pl->stream));
  return FFX_OK;
}

static int set_pairs_check_complete(ffx_plan* pl, int64_t* bad_pair) {
  ffx_index* ix = pl->idx;
  const int64_t P = pl->n_pairs;
  const int mode = pl->mode;
  unsigned long long bad = ~0ull;
  int32_t total = 0;
  memcpy(&bad, pl->check_back, sizeof bad);
  memcpy(&total, pl->check_back + 8, sizeof total);
  if (bad != ~0ull) {
    if (bad_pair) *bad_pair = (int64_t)bad;
    return fail(FFX_NOT_FOUND, "pair %llu names an ordinal with no rows in the index", bad);
  }
  if (pl->check_one_each) {
    pl->n_rows = P;
    if (!derive_kernel_exists(pl)) FFX_TRY(pl->desc.ensure(std::max<size_t>((size_t)P, 1) * sizeof(int2)));  // (else: on the run that needs a row list)
    pl->have_pairs = true;
    pl->derive_valid = false;
    pl->lazy_pairs = false;
    return FFX_OK;
  }
  // the row list is addressed with int32 and the counts were scanned in int32: where that scan could have
  // wrapped (a candidate contributes at most the rows of the longest document) the total is re-derived in int64
  if (mode != FFX_MODE_PASSAGE && mode != FFX_MODE_FIRSTP && P * ix->max_doc_rows > (int64_t)INT32_MAX) {
    size_t tmp = 0;
    auto wide = rocprim::make_transform_iterator(pl->cnt.as<int32_t>(), [] __device__(int32_t c) { return (long long)c; });
    FFX_TRY(pl->sort_tmp.ensure(64));
    long long* out64 = pl->sort_tmp.as<long long>();
    FFX_HIP(rocprim::reduce(nullptr, tmp, wide, out64, (long long)0, (size_t)P, rocprim::plus<long long>(), pl->stream));
    FFX_TRY(pl->scan_tmp.ensure(std::max<size_t>(tmp, 16)));
    FFX_HIP(rocprim::reduce(pl->scan_tmp.p, tmp, wide, out64, (long long)0, (size_t)P, rocprim::plus<long long>(), pl->stream));
    long long exact = 0;
    FFX_TRY(ffx_d2h(pl->device, &exact, out64, sizeof exact, pl->stream));
    FFX_HIP(hipStreamSynchronize(pl->stream));
    if (exact > (long long)INT32_MAX)
      return fail(FFX_BAD_ARG, "%lld gathered rows in one batch (more than 2^31-1): split the queries", exact);
  }
  if (total < 0) return fail(FFX_BAD_ARG, "more than 2^31-1 gathered rows in one batch: split the queries");
  pl->n_rows = total;
  const size_t rb = std::max<size_t>((size_t)total, 1);
  if (!derive_applies(pl)) FFX_TRY(pl->desc.ensure(rb * sizeof(int2)));  // (else on the run that needs it: 8 bytes per gathered row)
  if (pl->n_rows != pl->n_pairs) FFX_TRY(pl->rs.ensure(rb * sizeof(float)));
  pl->have_pairs = true;
  pl->derive_valid = false;
  pl->lazy_pairs = false;
  return FFX_OK;
}

static int finish_set_pairs(ffx_plan* pl, int64_t* bad_pair) {
  FFX_TRY(set_pairs_check_launch(pl));
  FFX_HIP(hipStreamSynchronize(pl->stream));
  return set_pairs_check_complete(pl, bad_pair);
}

// ffx_plan_set_pairs for candidate lists that already are in device memory (the sharded scorer routes a batch on the
// device): dev_pair_off / dev_pair_ids are copied device to device; host_pair_off is the same CSR offsets on the
// host (sizes, and the longest list for the ordering).  No first-stage scores (the owner side interpolates).
int ffx_plan_set_pairs_device(ffx_plan* pl, const int64_t* dev_pair_off, const int64_t* host_pair_off, int64_t n_queries,
                              const int32_t* dev_pair_ids, int mode, int64_t* bad_pair) {
  if (bad_pair) *bad_pair = -1;
  if (!pl || !dev_pair_off || !host_pair_off || n_queries < 0) return fail(FFX_BAD_ARG, "bad argument");
  if (mode < FFX_MODE_PASSAGE || mode > FFX_MODE_AVEP) return fail(FFX_BAD_ARG, "unknown mode %d", mode);
  if (n_queries > INT32_MAX - 1) return fail(FFX_BAD_ARG, "too many queries");
  if (host_pair_off[0] != 0) return fail(FFX_BAD_ARG, "pair_off[0] must be 0");
  for (int64_t q = 0; q < n_queries; ++q)
    if (host_pair_off[q + 1] < host_pair_off[q]) return fail(FFX_BAD_ARG, "pair_off not monotone at query %lld", (long long)q);
  const int64_t P = host_pair_off[n_queries];
  if (P > INT32_MAX - 1) return fail(FFX_BAD_ARG, "more than 2^31-2 pairs in one batch: split the queries");
  if (P > 0 && !dev_pair_ids) return fail(FFX_BAD_ARG, "pair_ids is NULL");
  ffx_index* ix = pl->idx;
  if ((mode == FFX_MODE_PASSAGE ? ix->n_psgs == 0 : ix->n_docs == 0) && P > 0) {
    if (bad_pair) *bad_pair = 0;
    return fail(FFX_NOT_FOUND, "the index holds no %s ids", mode == FFX_MODE_PASSAGE ? "passage" : "document");
  }
  FFX_HIP(hipSetDevice(ix->device));
  pl->have_pairs = false;
  pl->dense_given = false;
  pl->lazy_pairs = false;
  pl->ran = false;
  pl->n_queries = n_queries;
  pl->n_pairs = P;
  pl->mode = mode;
  pl->have_sparse = false;
  pl->host_pair_off.assign(host_pair_off, host_pair_off + n_queries + 1);
  pl->q_hint_valid = false;  // (the hint goes with the offsets it was made from: upload_query_hint makes a new one)
  const size_t pb = std::max<size_t>((size_t)P, 1);
  FFX_TRY(pl->pair_off.ensure((size_t)(n_queries + 1) * sizeof(int64_t)));
  FFX_TRY(pl->pair_ids.ensure(pb * sizeof(int32_t)));
  FFX_TRY(pl->cnt.ensure((pb + 1) * sizeof(int32_t)));
  FFX_TRY(pl->seg_off.ensure((pb + 1) * sizeof(int32_t)));
  FFX_TRY(pl->ff.ensure(pb * sizeof(float)));
  FFX_HIP(hipMemcpyAsync(pl->pair_off.p, dev_pair_off, (size_t)(n_queries + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, pl->stream));
  if (P > 0) FFX_HIP(hipMemcpyAsync(pl->pair_ids.p, dev_pair_ids, (size_t)P * sizeof(int32_t), hipMemcpyDeviceToDevice, pl->stream));
  pl->hstage_used = 0;
  return finish_set_pairs(pl, bad_pair);
}

// Two plans of one index that are to score the same candidate lists (the sharded scorer alternates between two plans
// as send buffers): the second one takes the validated lists of the first over device to device instead of
// uploading and validating them again.
int ffx_plan_copy_pairs(ffx_plan* dst, const ffx_plan* src) {
  if (!dst || !src || dst == src) return fail(FFX_BAD_ARG, "bad argument");
  if (dst->idx != src->idx) return fail(FFX_BAD_ARG, "the plans belong to different indexes");
  if (!src->have_pairs || src->lazy_pairs) return fail(FFX_BAD_ARG, "ffx_plan_set_pairs has not succeeded on the source plan");
  FFX_HIP(hipSetDevice(dst->device));
  const int64_t Q = src->n_queries, P = src->n_pairs;
  dst->have_pairs = false;
  dst->q_hint_valid = false;
  dst->lazy_pairs = false;
  dst->ran = false;
  const size_t pb = std::max<size_t>((size_t)P, 1);
  FFX_TRY(dst->pair_off.ensure((size_t)(Q + 1) * sizeof(int64_t)));
  FFX_TRY(dst->pair_ids.ensure(pb * sizeof(int32_t)));
  FFX_TRY(dst->cnt.ensure((pb + 1) * sizeof(int32_t)));
  FFX_TRY(dst->seg_off.ensure((pb + 1) * sizeof(int32_t)));
  FFX_TRY(dst->ff.ensure(pb * sizeof(float)));
  if (src->have_sparse) {
    FFX_TRY(dst->sparse.ensure(pb * sizeof(float)));
    FFX_TRY(dst->interp.ensure(pb * sizeof(float)));
  }
  const size_t rb = std::max<size_t>((size_t)src->n_rows, 1);
  if (src->desc.p) FFX_TRY(dst->desc.ensure(rb * sizeof(int2)));
  if (src->n_rows != P) FFX_TRY(dst->rs.ensure(rb * sizeof(float)));
  hipStream_t st = dst->stream;
  FFX_HIP(hipStreamSynchronize(src->stream));
  FFX_HIP(hipMemcpyAsync(dst->pair_off.p, src->pair_off.p, (size_t)(Q + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
  if (P > 0) {
    FFX_HIP(hipMemcpyAsync(dst->pair_ids.p, src->pair_ids.p, (size_t)P * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (src->have_sparse) FFX_HIP(hipMemcpyAsync(dst->sparse.p, src->sparse.p, (size_t)P * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
  FFX_HIP(hipMemcpyAsync(dst->cnt.p, src->cnt.p, (size_t)(P + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  FFX_HIP(hipMemcpyAsync(dst->seg_off.p, src->seg_off.p, (size_t)(P + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  FFX_HIP(hipStreamSynchronize(st));
  dst->n_queries = Q;
  dst->n_pairs = P;
  dst->n_rows = src->n_rows;
  dst->mode = src->mode;
  dst->have_sparse = src->have_sparse;
  dst->host_pair_off = src->host_pair_off;
  dst->q_hint_valid = false;  // (the hint goes with the offsets it was made from: upload_query_hint makes a new one)
  dst->order_base = src->order_base;
  dst->have_pairs = true;
  dst->derive_valid = false;
  return FFX_OK;
}

// Candidate lists for ffx_plan_early_stopping: uploaded, NOT validated.  An ordinal that names no
// row (e.g. -1 for an id the host could not resolve) is an error only if early stopping actually
// reaches it -- the reference looks ids up interval by interval (index/base.py:362-371).
int ffx_plan_set_pairs_lazy(ffx_plan* pl, const int64_t* pair_off, int64_t n_queries, const int32_t* pair_ids,
                            const float* sparse, int mode) {
  if (!pl || !pair_off || n_queries < 0) return fail(FFX_BAD_ARG, "bad argument");
  if (mode < FFX_MODE_PASSAGE || mode > FFX_MODE_AVEP) return fail(FFX_BAD_ARG, "unknown mode %d", mode);
  if (n_queries > INT32_MAX - 1) return fail(FFX_BAD_ARG, "too many queries");
  if (pair_off[0] != 0) return fail(FFX_BAD_ARG, "pair_off[0] must be 0");
  for (int64_t q = 0; q < n_queries; ++q)
    if (pair_off[q + 1] < pair_off[q]) return fail(FFX_BAD_ARG, "pair_off not monotone at query %lld", (long long)q);
  const int64_t P = pair_off[n_queries];
  if (P > INT32_MAX - 1) return fail(FFX_BAD_ARG, "more than 2^31-2 pairs in one batch: split the queries");
  if (P > 0 && (!pair_ids || !sparse)) return fail(FFX_BAD_ARG, "early stopping needs candidate ordinals and first-stage scores");
  FFX_HIP(hipSetDevice(pl->device));
  pl->have_pairs = false;
  pl->dense_given = false;
  pl->lazy_pairs = false;
  pl->ran = false;
  pl->n_queries = n_queries;
  pl->n_pairs = P;
  pl->n_rows = 0;
  pl->mode = mode;
  pl->have_sparse = true;
  pl->host_pair_off.assign(pair_off, pair_off + n_queries + 1);
  pl->q_hint_valid = false;  // (the hint goes with the offsets it was made from: upload_query_hint makes a new one)
  const size_t pb = std::max<size_t>((size_t)P, 1);
  FFX_TRY(pl->pair_off.ensure((size_t)(n_queries + 1) * sizeof(int64_t)));
  FFX_TRY(pl->pair_ids.ensure(pb * sizeof(int32_t)));
  FFX_TRY(pl->sparse.ensure(pb * sizeof(float)));
  FFX_TRY(pl->ff.ensure(pb * sizeof(float)));
  FFX_TRY(pl->interp.ensure(pb * sizeof(float)));
  pl->hstage_used = 0;
  FFX_TRY(plan_h2d(pl, pl->pair_off.p, pair_off, (size_t)(n_queries + 1) * sizeof(int64_t)));
  if (P > 0) {
    FFX_TRY(plan_h2d(pl, pl->pair_ids.p, pair_ids, (size_t)P * sizeof(int32_t)));
    FFX_TRY(plan_h2d(pl, pl->sparse.p, sparse, (size_t)P * sizeof(float)));
  }
  FFX_HIP(hipStreamSynchronize(pl->stream));
  pl->lazy_pairs = true;
  return FFX_OK;
}

}  // extern "C" (templates need C++ linkage)

// ---- ffx_internal.h: candidate lists written in place by the sharded scorer (comm.hip) ------------------------------
int ffx_plan_create_on(ffx_index* idx, hipStream_t stream, ffx_plan** out) {
  FFX_TRY(plan_create(idx, out, false));
  (*out)->stream = stream;
  return FFX_OK;
}

int ffx_plan_adopt_begin(ffx_plan* pl, int64_t n_queries, int64_t n_pairs, int mode, int64_t** d_pair_off, int32_t** d_pair_ids) {
  if (!pl || n_queries < 0 || n_pairs < 0 || !d_pair_off || !d_pair_ids) return fail(FFX_BAD_ARG, "bad argument");
  if (mode < FFX_MODE_PASSAGE || mode > FFX_MODE_AVEP) return fail(FFX_BAD_ARG, "unknown mode %d", mode);
  if (n_queries > INT32_MAX - 1) return fail(FFX_BAD_ARG, "too many queries");
  if (n_pairs > INT32_MAX - 1) return fail(FFX_BAD_ARG, "more than 2^31-2 pairs in one batch: split the queries");
  FFX_HIP(hipSetDevice(pl->device));
  pl->have_pairs = false;
  pl->dense_given = false;
  pl->lazy_pairs = false;
  pl->ran = false;
  pl->n_queries = n_queries;
  pl->n_pairs = n_pairs;
  pl->mode = mode;
  pl->have_sparse = false;
  pl->order_base = 0;
  const size_t pb = std::max<size_t>((size_t)n_pairs, 1);
  FFX_TRY(pl->pair_off.ensure((size_t)(n_queries + 1) * sizeof(int64_t)));
  FFX_TRY(pl->pair_ids.ensure(pb * sizeof(int32_t)));
  FFX_TRY(pl->cnt.ensure((pb + 1) * sizeof(int32_t)));
  FFX_TRY(pl->seg_off.ensure((pb + 1) * sizeof(int32_t)));
  FFX_TRY(pl->ff.ensure(pb * sizeof(float)));
  *d_pair_off = pl->pair_off.as<int64_t>();
  *d_pair_ids = pl->pair_ids.as<int32_t>();
  return FFX_OK;
}

int ffx_plan_adopt_check(ffx_plan* pl, const int64_t* host_pair_off, hipStream_t stream, const unsigned long long** d_verdict) {
  if (!pl || !host_pair_off || !d_verdict) return fail(FFX_BAD_ARG, "bad argument");
  if (host_pair_off[0] != 0 || host_pair_off[pl->n_queries] != pl->n_pairs) return fail(FFX_BAD_ARG, "the offsets do not describe the adopted lists");
  ffx_index* ix = pl->idx;
  pl->adopt_empty_index = (pl->mode == FFX_MODE_PASSAGE ? ix->n_psgs == 0 : ix->n_docs == 0) && pl->n_pairs > 0;
  FFX_HIP(hipSetDevice(pl->device));
  pl->host_pair_off.assign(host_pair_off, host_pair_off + pl->n_queries + 1);
  pl->q_hint_valid = false;  // (the hint goes with the offsets it was made from: upload_query_hint makes a new one)
  pl->hstage_used = 0;
  hipStream_t keep = pl->stream;
  pl->stream = stream;  // the validation belongs to the caller's stream, not to the one the plan scores on
  int rc = FFX_OK;
  if (pl->adopt_empty_index) {  // every candidate is unknown: the verdict is position 0
    const unsigned long long zero = 0;
    memcpy(pl->check_host, &zero, 8);
    pl->check_back = pl->check_host;
    if (hipMemsetAsync(pl->flags.p, 0, sizeof(unsigned long long), stream) != hipSuccess) rc = fail(FFX_HIP_ERROR, "memset failed");
  } else {
    rc = set_pairs_check_launch(pl);
  }
  pl->stream = keep;
  *d_verdict = pl->flags.as<unsigned long long>();
  return rc;
}

int ffx_plan_adopt_finish(ffx_plan* pl, hipStream_t stream, int64_t* bad_pair) {
  if (bad_pair) *bad_pair = -1;
  if (!pl) return fail(FFX_BAD_ARG, "bad argument");
  if (pl->adopt_empty_index) {
    if (bad_pair) *bad_pair = 0;
    return fail(FFX_NOT_FOUND, "the index holds no %s ids", pl->mode == FFX_MODE_PASSAGE ? "passage" : "document");
  }
  FFX_HIP(hipSetDevice(pl->device));
  hipStream_t keep = pl->stream;
  pl->stream = stream;
  const int rc = set_pairs_check_complete(pl, bad_pair);
  pl->stream = keep;
  return rc;
}

int ffx_plan_queries_room(ffx_plan* pl, int64_t n_queries, float** d_queries) {
  if (!pl || n_queries < 0 || !d_queries) return fail(FFX_BAD_ARG, "bad argument");
  if (n_queries > INT32_MAX) return fail(FFX_BAD_ARG, "too many queries");
  FFX_HIP(hipSetDevice(pl->device));
  FFX_TRY(pl->queries.ensure(std::max<size_t>((size_t)n_queries * pl->idx->qdim * sizeof(float), 16)));
  pl->n_queries_set = (int)n_queries;
  *d_queries = pl->queries.as<float>();
  return FFX_OK;
}

// what a row-deriving gather_dot reads instead of a row list (gather_dot.hip.h, Derive)
static ffx::Derive plan_derive(const ffx_plan* pl) {
  const ffx_index* ix = pl->idx;
  ffx::Derive dv;
  dv.tile_tab = pl->tile_tab.as<int2>();
  dv.seg_off = pl->seg_off.as<int32_t>();
  dv.pair_first = pl->pair_first.as<int32_t>();
  dv.doc_rows = ix->doc_rows_identity ? nullptr : ix->doc_rows.as<int32_t>();
  dv.pair_off = pl->pair_off.as<int64_t>();
  dv.n_pairs = (int)pl->n_pairs;
  dv.pair_ids = pl->pair_ids.as<int32_t>();
  dv.row_of = pl->mode == FFX_MODE_PASSAGE ? (ix->psgs_identity ? nullptr : ix->psg_row.as<int32_t>())
                                           : (ix->docs_identity ? nullptr : ix->doc_first.as<int32_t>());
  dv.n_queries = (int)pl->n_queries;
  dv.n_index_rows = (int)std::min<int64_t>(ix->capacity, INT32_MAX);
  dv.queries_per_pair = pl->n_pairs > 0 ? (double)pl->n_queries / (double)pl->n_pairs : 0.0;
  dv.q_hint = pl->q_hint_valid ? pl->q_hint.as<int32_t>() : nullptr;
  return dv;
}

// rows per tile of the row-deriving kernels (TL of launch_gather_dot_t; the tile table is made for it)
static int derive_tile(const ffx_plan* pl) {
  long long v = 0;
  if (ffx_experiment("derive_tile", &v) && (v == 8 || v == 16 || v == 32)) return (int)v;
  return pl->idx->dtype == FFX_DTYPE_F16 ? 16 : 8;
}

// rows of this storage type and length, in this mode, have a row-deriving kernel (either summation order: the fast order
// of d = 384 / 128 / 64 is served by the exact kernels)
static bool derive_kernel_exists(const ffx_plan* pl) {  // (one row per candidate: 411 - 417, 611 - 614)
  const ffx_index* ix = pl->idx;
  const int d = ix->qdim;
  return (ix->dtype == FFX_DTYPE_F32 || ix->dtype == FFX_DTYPE_F16) && !pl->ov_dim && !ffx_debug_variant("no_derive", nullptr) &&
         (d == 768 || d == 1024 || d == 512 || d == 384 || d == 256 || d == 128 || d == 64);
}
static bool derive_applies(const ffx_plan* pl) {  // (candidates of several rows: 311 / 312, 511 / 512)
  return derive_kernel_exists(pl) && (pl->idx->qdim == 768 || pl->idx->qdim == 1024) && (pl->mode == FFX_MODE_MAXP || pl->mode == FFX_MODE_AVEP);
}

// pick and launch the gather_dot instantiation for rows of type T (float / _Float16); the
// variant id lands in ffx_stats.kernel_variant (+1000 for half rows)
template <typename T>
static int launch_gather_dot_t(ffx_plan* pl, bool exact, float* rs) {
  const ffx_index* ix = pl->idx;
  hipStream_t st = pl->stream;
  if (!pl->ov_dim) {
    // d = S * b with S in {2, 4, 8} and b a specialised row length (1536 = 2*768, 2048 = 2*1024,
    // 3072 = 4*768, 4096 = 4*1024, ...): score the S*N b-element rows the same memory holds and
    // combine the S partial sums pairwise -- NumPy's own recursion (aux_kernels.hip.h)
    static const int bases[] = {1024, 768, 512, 384, 256, 128, 64};
    int base = 0, S = 0;
    bool own_kernel = false;
    for (int b : bases) own_kernel = own_kernel || ix->dim == b;
    for (int b : bases)
      for (int m = 2; m <= 8 && !base && !own_kernel; m *= 2)
        if (ix->dim == b * m) base = b, S = m;
    constexpr int B = 16;
    const long n_virtual = ((pl->n_rows + B - 1) / B) * B * (long)S;
    if (base && n_virtual <= INT32_MAX && ix->capacity * S <= INT32_MAX && (long)pl->n_queries * S <= INT32_MAX) {
      FFX_TRY(pl->vdesc.ensure((size_t)n_virtual * sizeof(int2)));
      FFX_TRY(pl->rs_v.ensure((size_t)n_virtual * sizeof(float)));
      ffx::segment_desc_kernel<<<blocks_for(n_virtual, 256), 256, 0, st>>>(pl->desc.as<int2>(), (int)pl->n_rows, S, B,
                                                                          pl->vdesc.as<int2>(), n_virtual);
      FFX_HIP(hipGetLastError());
      pl->ov_dim = base;
      pl->ov_rows = (int)n_virtual;
      const int rc = launch_gather_dot_t<T>(pl, exact, pl->rs_v.as<float>());
      pl->ov_dim = pl->ov_rows = 0;
      FFX_TRY(rc);
      ffx::combine_segments_kernel<<<blocks_for(pl->n_rows, 256), 256, 0, st>>>(pl->rs_v.as<float>(), (int)pl->n_rows, S, B, rs);
      FFX_HIP(hipGetLastError());
      pl->variant += 10000 * S;  // e.g. 20111: two 768-element segments through variant 111
      return FFX_OK;
    }
  }
  const int d = pl->ov_dim ? pl->ov_dim : ix->dim;
  const int R = pl->ov_dim ? pl->ov_rows : (int)pl->n_rows;
  const T* V = static_cast<const T*>(ix->vectors);
  const float* Q = pl->queries.as<float>();
  const int2* desc = pl->ov_dim ? pl->vdesc.as<int2>() : pl->desc.as<int2>();
  constexpr int HALF = sizeof(T) == 2 ? 1000 : 0;
  // 4 rows in flight per wave for both row types (8 is slower: tools/kernel_ab.py,
  // profiles/r01/kernel_ab_f16.log); half rows run best with 16-row tiles, float rows with 8
  constexpr int UL = 4;
  constexpr int TL = sizeof(T) == 2 ? 16 : 8;
#define FFX_EXACT(LEAVES, NPER, U, ID)                                                            \
  do {                                                                                            \
    constexpr int RPW_ = 64 / (8 * LEAVES);                                                       \
    constexpr int PASSES_ = RPW_ == 1 ? 16 : ((64 / RPW_) < 16 ? (64 / RPW_) : 16);               \
    constexpr int TILE_ = PASSES_ * RPW_;                                                         \
    ffx::gather_dot_exact_kernel<T, LEAVES, NPER, U><<<blocks_for(R, 4 * TILE_), 256, 0, st>>>(V, Q, desc, rs, R); \
    pl->variant = ID + HALF;                                                                      \
  } while (0)
#ifdef FFX_EXPERIMENTS
  // tuning hook (tools/kernel_ab.py; experiments build only): FFX_DEBUG_VARIANT=gather=N selects an experimental d=768 instantiation
  long long exp_value = 0;
  const int exp_variant = ffx_experiment("gather", &exp_value) ? (int)exp_value : 0;  // tuning hook (tools/kernel_ab.py)
  if (exp_variant >= 800 && exp_variant < 900) {  // the strided-load form of the small-row dims, for A/B
    switch (d) {
      case 512: FFX_EXACT(4, 16, 2, 803); break;
      case 384: FFX_EXACT(4, 12, 4, 804); break;
      case 256: FFX_EXACT(2, 16, 2, 805); break;
      case 128: FFX_EXACT(1, 16, 2, 806); break;
      case 64: FFX_EXACT(1, 8, 4, 807); break;
      default: return fail(FFX_BAD_ARG, "no strided variant for dim %d", d);
    }
    FFX_HIP(hipGetLastError());
    return FFX_OK;
  }
  if (d == 768 && exp_variant >= 900) {
    const unsigned waves = (unsigned)R;
    switch (exp_variant) {
#define FFX_EXP_EXACT(ID, U, PASSES, NT) \
  case ID: ffx::gather_dot_exact_kernel<T, 8, 12, U, PASSES, NT><<<blocks_for(waves, 4 * PASSES), 256, 0, st>>>(V, Q, desc, rs, R); break
      FFX_EXP_EXACT(900, 4, 16, false);  // round-1 first shipping form (strided element loads)
      FFX_EXP_EXACT(903, 4, 16, true);
      FFX_EXP_EXACT(907, 4, 8, false);
#undef FFX_EXP_EXACT
#define FFX_EXP_LDS(ID, U, TILE, NT) \
  case ID: ffx::gather_dot_exact_lds_kernel<T, 3, U, TILE, NT><<<blocks_for(waves, 4 * TILE), 256, 0, st>>>(V, Q, desc, rs, R, ffx::Derive{}); break
      FFX_EXP_LDS(911, 4, 16, true);
      FFX_EXP_LDS(912, 2, 8, true);
      FFX_EXP_LDS(913, 4, 8, true);
      FFX_EXP_LDS(914, 4, 16, false);
      FFX_EXP_LDS(916, 8, 16, true);
      FFX_EXP_LDS(919, 8, 8, true);
      FFX_EXP_LDS(920, 16, 16, true);
#undef FFX_EXP_LDS
      // the shipping kernel as a persistent grid (experiment, see gather_dot.hip.h): strided tiles / contiguous shares, 6 or 12 workgroups per CU
      case 921: ffx::gather_dot_exact_lds_persistent_kernel<T, 3, 4, 8, true, false><<<256 * 6, 256, 0, st>>>(V, Q, desc, rs, R); break;
      case 922: ffx::gather_dot_exact_lds_persistent_kernel<T, 3, 4, 8, true, true><<<256 * 6, 256, 0, st>>>(V, Q, desc, rs, R); break;
      case 923: ffx::gather_dot_exact_lds_persistent_kernel<T, 3, 4, 8, true, false><<<256 * 12, 256, 0, st>>>(V, Q, desc, rs, R); break;
      case 924: ffx::gather_dot_exact_lds_persistent_kernel<T, 3, 4, 16, true, true><<<256 * 6, 256, 0, st>>>(V, Q, desc, rs, R); break;
      case 925: ffx::gather_dot_exact_lds_kernel<T, 3, 4, 8, true><<<blocks_for(waves, 2 * 8), 128, 0, st>>>(V, Q, desc, rs, R, ffx::Derive{}); break;  // workgroups of two waves
      case 951: ffx::gather_dot_fast_kernel<T, 3, 4, 16, true><<<blocks_for(R, 64), 256, 0, st>>>(V, Q, desc, rs, R, ffx::Derive{}); break;
      case 953: ffx::gather_dot_fast_kernel<T, 3, 8, 32, true><<<blocks_for(R, 128), 256, 0, st>>>(V, Q, desc, rs, R, ffx::Derive{}); break;
      case 955: ffx::gather_dot_fast_kernel<T, 3, 4, 8, true><<<blocks_for(R, 32), 256, 0, st>>>(V, Q, desc, rs, R, ffx::Derive{}); break;
      case 958: ffx::gather_dot_fast_kernel<T, 3, 8, 8, true><<<blocks_for(R, 32), 256, 0, st>>>(V, Q, desc, rs, R, ffx::Derive{}); break;
      case 959: ffx::gather_dot_fast_kernel<T, 3, 16, 16, true><<<blocks_for(R, 64), 256, 0, st>>>(V, Q, desc, rs, R, ffx::Derive{}); break;
      default: return fail(FFX_BAD_ARG, "unknown gather variant %d (FFX_DEBUG_VARIANT)", exp_variant);
    }
    pl->variant = exp_variant;
    FFX_HIP(hipGetLastError());
    return FFX_OK;
  }
#endif  // FFX_EXPERIMENTS
  // d = 768 / 1024 (8 leaves): full-line non-temporal loads + LDS transpose (16.0 ms vs 17.7 ms for
  // the strided form at 8.8M x 768 fp32, k=5000: tools/kernel_ab.py, profiles/r01/kernel_ab_2.log)
  if (exact) {
    switch (pl->forced_pieces && !pl->segments.empty() ? -1 : d) {
      case 768:
        if (pl->use_derive) {  // no row list: the waves work their rows out of the candidate lists (311 / 1311)
#ifdef FFX_EXPERIMENTS
          if (derive_tile(pl) == 16 && TL != 16) {
            ffx::gather_dot_exact_lds_kernel<T, 3, UL, 16, true, 1><<<blocks_for(R, 4 * 16), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));
          } else if (derive_tile(pl) == 32) {
            ffx::gather_dot_exact_lds_kernel<T, 3, UL, 32, true, 1><<<blocks_for(R, 4 * 32), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));
          } else if (derive_tile(pl) == 8 && TL != 8) {
            ffx::gather_dot_exact_lds_kernel<T, 3, UL, 8, true, 1><<<blocks_for(R, 4 * 8), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));
          } else
#endif
          ffx::gather_dot_exact_lds_kernel<T, 3, UL, TL, true, 1><<<blocks_for(R, 4 * TL), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));
        } else if (pl->use_derive1) {  // one row per candidate, no row list either (411 / 1411)
#ifdef FFX_EXPERIMENTS
          long long tv = 0, uv = 0;
          (void)ffx_experiment("derive_tile", &tv);
          (void)ffx_experiment("derive_u", &uv);
          if (tv == 16 && uv == 0) {
            ffx::gather_dot_exact_lds_kernel<T, 3, UL, 16, true, 2><<<blocks_for(R, 4 * 16), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));
          } else if (tv == 32) {
            ffx::gather_dot_exact_lds_kernel<T, 3, UL, 32, true, 2><<<blocks_for(R, 4 * 32), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));
          } else if (tv == 16 && uv == 8) {
            ffx::gather_dot_exact_lds_kernel<T, 3, 8, 16, true, 2><<<blocks_for(R, 4 * 16), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));
          } else if (tv == 8 && uv == 8) {
            ffx::gather_dot_exact_lds_kernel<T, 3, 8, 8, true, 2><<<blocks_for(R, 4 * 8), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));
          } else if (tv == 8 && uv == 2) {
            ffx::gather_dot_exact_lds_kernel<T, 3, 2, 8, true, 2><<<blocks_for(R, 4 * 8), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));
          } else if (tv == 4) {
            ffx::gather_dot_exact_lds_kernel<T, 3, 4, 4, true, 2><<<blocks_for(R, 4 * 4), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));
          } else
#endif
          ffx::gather_dot_exact_lds_kernel<T, 3, UL, TL, true, 2><<<blocks_for(R, 4 * TL), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));
        } else
          ffx::gather_dot_exact_lds_kernel<T, 3, UL, TL, true><<<blocks_for(R, 4 * TL), 256, 0, st>>>(V, Q, desc, rs, R, ffx::Derive{});
        pl->variant = (pl->use_derive ? 311 : pl->use_derive1 ? 411 : 111) + HALF;
        break;
      case 1024:
        if (pl->use_derive)
          ffx::gather_dot_exact_lds_kernel<T, 4, UL / 2, TL, true, 1><<<blocks_for(R, 4 * TL), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));
        else if (pl->use_derive1)
          ffx::gather_dot_exact_lds_kernel<T, 4, UL / 2, TL, true, 2><<<blocks_for(R, 4 * TL), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));
        else
          ffx::gather_dot_exact_lds_kernel<T, 4, UL / 2, TL, true><<<blocks_for(R, 4 * TL), 256, 0, st>>>(V, Q, desc, rs, R, ffx::Derive{});
        pl->variant = (pl->use_derive ? 312 : pl->use_derive1 ? 412 : 112) + HALF;
        break;
#define FFX_MULTI(LEAVES, NPER, U, TILE, ID)                                                                        \
  do {                                                                                                              \
    if (pl->use_derive1)                                                                                            \
      ffx::gather_dot_exact_lds_multi_kernel<T, LEAVES, NPER, U, TILE, true, 2><<<blocks_for(R, 4 * TILE), 256, 0, st>>>( \
          V, Q, nullptr, rs, R, plan_derive(pl));                                                                   \
    else                                                                                                            \
      ffx::gather_dot_exact_lds_multi_kernel<T, LEAVES, NPER, U, TILE, true><<<blocks_for(R, 4 * TILE), 256, 0, st>>>( \
          V, Q, desc, rs, R, ffx::Derive{});                                                                        \
    pl->variant = ID + (pl->use_derive1 ? 300 : 0) + HALF;                                                          \
  } while (0)
      // smaller rows: several rows share a wave, same full-line nt loads + LDS transpose
      case 512: FFX_MULTI(4, 16, 2, 16, 113); break;
      case 384: FFX_MULTI(4, 12, 2, 16, 114); break;
      case 256: FFX_MULTI(2, 16, 2, 32, 115); break;
      case 128: FFX_MULTI(1, 16, 2, 64, 116); break;
      case 64: FFX_MULTI(1, 8, 4, 64, 117); break;
#undef FFX_MULTI
      default: {
        if (pl->n_leaves > 8192) return fail(FFX_BAD_ARG, "dim %d too large for the exact kernel", d);
        // any other row length with at most 8 leaves (8 <= d <= 1024): the strided form with a run-time leaf
        // table -- every lane loads what it sums, nothing is staged or combined serially
        // the any-d kernel on one segment of the row: SEG_LEN elements from column COL0 on, leaf table LS (relative to the
        // segment), NL leaves, combine steps CS, row sums to OUT
#define FFX_ANYD(LG, U, TAILS_, COL0, LS, NL, CS, OUT)                                                                        \
  do {                                                                                                                        \
    constexpr int TILE_ = ffx::exact_default_passes(LG) * (64 / (8 * LG));                                                    \
    if (TAILS_)                                                                                                               \
      ffx::gather_dot_exact_anyd_kernel<T, LG, U, true><<<blocks_for(R, 4 * TILE_), 256, 0, st>>>(V, Q, desc, OUT, R, d, COL0, LS, NL, CS);  \
    else                                                                                                                      \
      ffx::gather_dot_exact_anyd_kernel<T, LG, U, false><<<blocks_for(R, 4 * TILE_), 256, 0, st>>>(V, Q, desc, OUT, R, d, COL0, LS, NL, CS); \
  } while (0)
#define FFX_ANYD_LG(LG, U_SEL, TAILS_, COL0, LS, NL, CS, OUT)            \
  do {                                                                   \
    if (U_SEL == 2) FFX_ANYD(LG, 2, TAILS_, COL0, LS, NL, CS, OUT);      \
    else if (U_SEL == 8) FFX_ANYD(LG, 8, TAILS_, COL0, LS, NL, CS, OUT); \
    else FFX_ANYD(LG, 4, TAILS_, COL0, LS, NL, CS, OUT);                 \
  } while (0)
#define FFX_ANYD_ANY(U_SEL, TAILS_, COL0, LS, NL, CS, OUT)                     \
  do {                                                                         \
    if (NL == 1) FFX_ANYD_LG(1, U_SEL, TAILS_, COL0, LS, NL, CS, OUT);         \
    else if (NL == 2) FFX_ANYD_LG(2, U_SEL, TAILS_, COL0, LS, NL, CS, OUT);    \
    else if (NL <= 4) FFX_ANYD_LG(4, U_SEL, TAILS_, COL0, LS, NL, CS, OUT);    \
    else FFX_ANYD_LG(8, U_SEL, TAILS_, COL0, LS, NL, CS, OUT);                 \
  } while (0)
        long long u_value = 0;  // A/B hook (tools/dims_bench.py): passes of loads in flight per wave
        const bool u_env = ffx_experiment("anyd_u", &u_value);
        if (pl->n_leaves <= 8 && d >= 8 && pl->anyd_ok && !pl->forced_pieces && !ffx_experiment("generic_staged", nullptr)) {
          const int* LS = pl->leaf_start.as<int>();
          const int* CS = pl->combine_steps.as<int>();
          const int nl = pl->n_leaves;
          // measured (profiles/r02/anyd_kernel_dims.json): 2 passes in flight win for 3 .. 7 leaves (d = 300: 5.09 vs
          // 4.59 TB/s, d = 520: 5.61 vs 5.20), 4 for one or two leaves and for long rows (d = 1000: 5.85 vs 5.65)
          const int u_sel = u_env ? (int)u_value : ((nl >= 3 && d < 768) ? 2 : 4);
          FFX_ANYD_ANY(u_sel, pl->anyd_tails, 0, LS, nl, CS, rs);
          pl->variant = 130 + nl + HALF;
          break;
        }
        // more than 8 leaves (d > 1024 that is not S x a specialised length: 1100, 1280, 2000, 5000 ...): the row is
        // cut along NumPy's recursion into segments of <= 8 leaves (ffx_plan_create), every segment is summed by the
        // any-d kernel (one launch per segment over all rows: a segment is >= 2 KB of consecutive bytes of a row),
        // and combine_tree_kernel adds the segment sums in the order of the recursion.  Bit-identical to NumPy; the
        // staged generic kernel it replaces walked every row out of LDS with one wave (1 - 3.5 TB/s).
        if (!pl->ov_dim && !pl->segments.empty() && !ffx_experiment("generic_staged", nullptr)) {
          const int G = (int)pl->segments.size();
          FFX_TRY(pl->rs_v.ensure((size_t)G * (size_t)R * sizeof(float)));
          const int* tab = pl->seg_tables.as<int>();
          for (int g = 0; g < G; ++g) {
            const ffx_plan::Segment& sg = pl->segments[(size_t)g];
            const int* LS = tab + sg.table_at;
            const int* CS = tab + sg.steps_at;
            const int nl = sg.n_leaves;
            const int u_sel = u_env ? (int)u_value : ((nl >= 3 && sg.len < 768) ? 2 : 4);
            float* out = pl->rs_v.as<float>() + (size_t)g * R;
            FFX_ANYD_ANY(u_sel, sg.tails, sg.col0, LS, nl, CS, out);
            FFX_HIP(hipGetLastError());
          }
          ffx::combine_tree_kernel<<<blocks_for(R, 256), 256, (size_t)G * 256 * sizeof(float), st>>>(pl->rs_v.as<float>(), R, G, tab + pl->seg_steps_at,
                                                                                                 pl->seg_n_steps, rs);
          pl->variant = 140 + HALF + 10000 * G;  // e.g. 20140: two segments through the any-d kernel
          break;
        }
        // any other row length: the row is staged in LDS (coalesced, independent loads) and the
        // leaves are walked from there; rows too long for a wave's share of LDS use the plain form
        const int dp = (d + 3) & ~3, lp = (pl->n_leaves + 3) & ~3;
        int RP = 1;  // rows a wave takes per pass: 8 / (leaves per row rounded up to a power of two)
        while (RP < 8 && pl->n_leaves * (RP * 2) <= 8) RP *= 2;
        while (RP > 1 && (size_t)RP * (2 * dp + lp) * sizeof(float) > (12 << 10)) RP /= 2;  // keep 4 waves per block
        const size_t per_wave = (size_t)RP * (2 * dp + lp) * sizeof(float);
        const int waves = (int)std::min<size_t>(4, (48 << 10) / per_wave);
        if (waves >= 1 && !ffx_experiment("generic_plain", nullptr)) {
          const int rows_per_wave = 8 * RP;
          ffx::gather_dot_exact_generic_staged_kernel<T><<<blocks_for(R, waves * rows_per_wave), 64 * waves, per_wave * waves, st>>>(
              V, Q, desc, rs, R, d, pl->leaf_start.as<int>(), pl->n_leaves, pl->combine_steps.as<int>(), dp, lp, rows_per_wave, RP);
          pl->variant = 198 + HALF;
          break;
        }
        ffx::gather_dot_exact_generic_kernel<T><<<(unsigned)R, 64, (size_t)pl->n_leaves * sizeof(float), st>>>(
            V, Q, desc, rs, R, d, pl->leaf_start.as<int>(), pl->n_leaves, pl->combine_steps.as<int>());
        pl->variant = 199 + HALF;
      }
    }
  } else {
    switch (d) {
      // row list: 211 - 214; rows derived in the kernel, as in the exact kernels: several per candidate 511 / 512, one 611 - 614
#define FFX_FAST(NCH, U, ID)                                                                                                            \
  do {                                                                                                                                  \
    if (pl->use_derive) {                                                                                                               \
      ffx::gather_dot_fast_kernel<T, NCH, U, TL, true, 1><<<blocks_for(R, 4 * TL), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));  \
      pl->variant = ID + 300 + HALF;                                                                                                    \
    } else if (pl->use_derive1) {                                                                                                       \
      ffx::gather_dot_fast_kernel<T, NCH, U, TL, true, 2><<<blocks_for(R, 4 * TL), 256, 0, st>>>(V, Q, nullptr, rs, R, plan_derive(pl));  \
      pl->variant = ID + 400 + HALF;                                                                                                    \
    } else {                                                                                                                            \
      ffx::gather_dot_fast_kernel<T, NCH, U, TL, true><<<blocks_for(R, 4 * TL), 256, 0, st>>>(V, Q, desc, rs, R, ffx::Derive{});         \
      pl->variant = ID + HALF;                                                                                                          \
    }                                                                                                                                   \
  } while (0)
      case 768: FFX_FAST(3, UL, 211); break;
      case 1024: FFX_FAST(4, UL / 2, 212); break;
      case 512: FFX_FAST(2, UL, 213); break;  // (use_derive is never set here: derive_applies)
      case 256: FFX_FAST(1, UL, 214); break;
#undef FFX_FAST
      default:
        // dims with a multi-row exact kernel but no FMA specialisation: the exact kernel is HBM-bound
        // too (and 1.3-2.8x faster than the one-row-per-wave generic kernel), so "fast" uses it
        if (d == 384 || d == 128 || d == 64) return launch_gather_dot_t<T>(pl, true, rs);
        if (d % 4 == 0) {
          ffx::gather_dot_fast_generic_kernel<T><<<blocks_for(R, 4), 256, 0, st>>>(V, Q, desc, rs, R, d);
          pl->variant = 298 + HALF;
        } else {  // rows are not 8/16-byte aligned: the exact generic kernels are the only scalar path
          return launch_gather_dot_t<T>(pl, true, rs);
        }
    }
  }
#undef FFX_ANYD_ANY
#undef FFX_ANYD_LG
#undef FFX_ANYD
#undef FFX_EXACT
  FFX_HIP(hipGetLastError());
  return FFX_OK;
}

// product-quantized rows: decode-then-dot (exact, variants 21xx) or ADC (fast, 2300)
// subspaces == leaves of NumPy's tree (every leaf has length Ds, M in {1,2,4,8,16}): exact ADC applies
// (FFX_DEBUG_VARIANT=pq_no_adc forces decode-then-dot: tests and A/B)
static bool pq_exact_adc_applies(const ffx_plan* pl) {
  const ffx_index* ix = pl->idx;
  if (ix->dtype != FFX_DTYPE_PQ8) return false;
  const int d = ix->qdim, M = ix->dim, Ds = ix->pq_ds;
  bool aligned = !ffx_debug_variant("pq_no_adc", nullptr) && Ds <= 128 && pl->n_leaves == M && (M == 1 || M == 2 || M == 4 || M == 8 || M == 16);
  if (aligned) {
    std::vector<int> bounds;
    numpy_leaf_plan(0, d, bounds);
    for (int m = 0; m < M && aligned; ++m) aligned = bounds[(size_t)m] == m * Ds;
  }
  return aligned;
}

static int64_t longest_segment(const int64_t* off, int64_t n);

// single-row candidates on exact-ADC rows: expand + look-ups + interpolation fused (pq_adc_exact_pairs_kernel); with
// *sorted != NULL and queries of 129 .. 8192 candidates also the per-query ordering (pq_adc_exact_pairs_sort_kernel: a
// workgroup per query), and *sorted says whether that happened
static int launch_pq_fused(ffx_plan* pl, float a, float b, uint32_t flags = 0, bool* sorted = nullptr) {
  const ffx_index* ix = pl->idx;
  const int M = ix->dim, Ks = ix->pq_ks, Ds = ix->pq_ds;
  const int64_t P = pl->n_pairs;
  hipStream_t st = pl->stream;
  const size_t lut_bytes = (size_t)pl->n_queries * M * Ks * sizeof(float);
  FFX_TRY(pl->lut.ensure(std::max<size_t>(lut_bytes, 16)));
  if (pl->n_queries > 0) {
    ffx::pq_lut_exact_kernel<<<dim3((unsigned)pl->n_queries, (unsigned)M), 256, 0, st>>>(pl->queries.as<float>(), ix->codebook_t.as<float>(), M, Ks, Ds,
                                                                                        pl->lut.as<float>());
    FFX_HIP(hipGetLastError());
  }
  const int32_t* row_of = pl->mode == FFX_MODE_PASSAGE ? (ix->psgs_identity ? nullptr : ix->psg_row.as<int32_t>())
                                                       : (ix->docs_identity ? nullptr : ix->doc_first.as<int32_t>());
  const size_t lds = (size_t)M * Ks * sizeof(float);
  const double qpp = (double)pl->n_queries / (double)P;
  if (sorted) *sorted = false;
  if (sorted && (M == 8 || M == 16) && lds <= (32u << 10) && !ffx_debug_variant("pq_sort_apart", nullptr)) {
    const int64_t longest = longest_segment(pl->host_pair_off.data(), pl->n_queries);
    if (longest >= 129 && longest <= 8192) {
      FFX_TRY(pl->order.ensure((size_t)P * sizeof(int32_t)));
      const unsigned nq = (unsigned)pl->n_queries;
      const int sort_ff = (flags & FFX_FLAG_SORT_FF) ? 1 : 0;
#define FFX_ADCS(MM, BS, IPT)                                                                                                                     \
  ffx::pq_adc_exact_pairs_sort_kernel<MM, BS, IPT><<<nq, BS, lds, st>>>(static_cast<const uint8_t*>(ix->vectors), Ks, pl->lut.as<float>(),        \
      pl->pair_ids.as<int32_t>(), pl->pair_off.as<int64_t>(), row_of, pl->have_sparse ? pl->sparse.as<float>() : nullptr, a, b, sort_ff,          \
      pl->ff.as<float>(), pl->interp.as<float>(), pl->order_base, pl->order.as<int32_t>(), (long)ix->capacity)
#define FFX_ADCS_M(MM)                               \
  do {                                               \
    if (longest <= 1024) FFX_ADCS(MM, 256, 4);       \
    else if (longest <= 4096) FFX_ADCS(MM, 512, 8);  \
    else if (longest <= 5120) FFX_ADCS(MM, 512, 10); \
    else FFX_ADCS(MM, 1024, 8);                      \
  } while (0)
      if (M == 8) FFX_ADCS_M(8); else FFX_ADCS_M(16);
#undef FFX_ADCS_M
#undef FFX_ADCS
      FFX_HIP(hipGetLastError());
      pl->variant = 2500 + M;
      *sorted = true;
      return FFX_OK;
    }
  }
#define FFX_ADCP(MM)                                                                                                               \
  ffx::pq_adc_exact_pairs_kernel<MM><<<blocks_for(P, 256), 256, lds, st>>>(static_cast<const uint8_t*>(ix->vectors), Ks, pl->lut.as<float>(), \
      pl->pair_ids.as<int32_t>(), pl->pair_off.as<int64_t>(), (int)pl->n_queries, qpp, row_of,                                    \
      pl->have_sparse ? pl->sparse.as<float>() : nullptr, a, b, pl->ff.as<float>(), pl->interp.as<float>(), (long)P, (long)ix->capacity)
  switch (M) {
    case 1: FFX_ADCP(1); break;
    case 2: FFX_ADCP(2); break;
    case 4: FFX_ADCP(4); break;
    case 8: FFX_ADCP(8); break;
    default: FFX_ADCP(16); break;
  }
#undef FFX_ADCP
  FFX_HIP(hipGetLastError());
  pl->variant = 2400 + M;
  return FFX_OK;
}

static int launch_gather_dot_pq(ffx_plan* pl, bool exact, float* rs) {
  const ffx_index* ix = pl->idx;
  const int d = ix->qdim, M = ix->dim, Ks = ix->pq_ks, Ds = ix->pq_ds;
  const int R = (int)pl->n_rows;
  const uint8_t* codes = static_cast<const uint8_t*>(ix->vectors);
  const float* cb = ix->codebook.as<float>();
  const float* Q = pl->queries.as<float>();
  const int2* desc = pl->desc.as<int2>();
  hipStream_t st = pl->stream;
  // (the exact table is also the fastest one, so exact_sum=false takes it too when it applies)
  const bool aligned = pq_exact_adc_applies(pl);
  if (aligned) {
    const size_t lut_bytes = (size_t)pl->n_queries * M * Ks * sizeof(float);
    FFX_TRY(pl->lut.ensure(std::max<size_t>(lut_bytes, 16)));
    if (pl->n_queries > 0) {
      ffx::pq_lut_exact_kernel<<<dim3((unsigned)pl->n_queries, (unsigned)M), 256, 0, st>>>(Q, ix->codebook_t.as<float>(), M, Ks, Ds,
                                                                                          pl->lut.as<float>());
      FFX_HIP(hipGetLastError());
    }
    const size_t lds = (size_t)M * Ks * sizeof(float);
#define FFX_ADCX(MM) ffx::pq_adc_exact_kernel<MM><<<blocks_for(R, 256), 256, lds, st>>>(codes, Ks, pl->lut.as<float>(), desc, rs, R)
    switch (M) {
      case 1: FFX_ADCX(1); break;
      case 2: FFX_ADCX(2); break;
      case 4: FFX_ADCX(4); break;
      case 8: FFX_ADCX(8); break;
      default: FFX_ADCX(16); break;
    }
#undef FFX_ADCX
    pl->variant = 2200 + M;
  } else if (exact) {
    const bool lds_ok = (d == 768 || d == 1024) && Ds % 4 == 0 && (M == 8 || M == 16 || M == 32);
    if (lds_ok) {
#define FFX_PQ(NCH, MW) ffx::gather_dot_pq_lds_kernel<NCH, MW, 4, 16><<<blocks_for(R, 64), 256, 0, st>>>(codes, cb, Ks, Ds, Q, desc, rs, R)
      if (d == 768) {
        if (M == 8) FFX_PQ(3, 1); else if (M == 16) FFX_PQ(3, 2); else FFX_PQ(3, 4);
      } else {
        if (M == 8) FFX_PQ(4, 1); else if (M == 16) FFX_PQ(4, 2); else FFX_PQ(4, 4);
      }
#undef FFX_PQ
      pl->variant = 2100 + M;
    } else {
      if (pl->n_leaves > 8192) return fail(FFX_BAD_ARG, "dim %d too large for the exact kernel", d);
      ffx::gather_dot_pq_exact_generic_kernel<<<(unsigned)R, 64, (size_t)pl->n_leaves * sizeof(float), st>>>(
          codes, M, cb, Ks, Ds, Q, desc, rs, R, d, pl->leaf_start.as<int>(), pl->n_leaves, pl->combine_steps.as<int>());
      pl->variant = 2199;
    }
  } else {
    const size_t lut_bytes = (size_t)pl->n_queries * M * Ks * sizeof(float);
    FFX_TRY(pl->lut.ensure(std::max<size_t>(lut_bytes, 16)));
    if (pl->n_queries > 0) {
      ffx::pq_lut_kernel<<<dim3((unsigned)pl->n_queries, (unsigned)M), 256, 0, st>>>(Q, ix->codebook_t.as<float>(), M, Ks, Ds,
                                                                                    pl->lut.as<float>());
      FFX_HIP(hipGetLastError());
    }
    // the block's table lives in LDS when it fits: 64 KB by default, up to the CU's 160 KB on request;
    // a larger table (NanoPQ(M=96, Ks=256) on d=768 is 96 KB; M=192 would be 192 KB) is read through L2
    size_t lds = (size_t)M * Ks * sizeof(float);
    int stage = 1;
    if (lds > 64 * 1024) {
      const bool raise_ok = lds <= 160 * 1024 &&
          hipFuncSetAttribute(reinterpret_cast<const void*>(ffx::pq_adc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
      if (!raise_ok) {
        (void)hipGetLastError();
        stage = 0;
        lds = 0;
      }
    }
    ffx::pq_adc_kernel<<<blocks_for(R, 256), 256, lds, st>>>(codes, M, Ks, pl->lut.as<float>(), desc, rs, R, stage);
    pl->variant = stage ? 2300 : 2301;
  }
  FFX_HIP(hipGetLastError());
  return FFX_OK;
}

static int launch_gather_dot(ffx_plan* pl, bool exact, float* rs) {
  switch (pl->idx->dtype) {
    case FFX_DTYPE_F16: return launch_gather_dot_t<_Float16>(pl, exact, rs);
    case FFX_DTYPE_PQ8: return launch_gather_dot_pq(pl, exact, rs);
    default: return launch_gather_dot_t<float>(pl, exact, rs);
  }
}

extern "C" {

static int64_t longest_segment(const int64_t* off, int64_t n) {
  int64_t m = 0;
  for (int64_t q = 0; q < n; ++q) m = std::max(m, off[q + 1] - off[q]);
  return m;
}

// per-query order: key descending, ties by position (the radix sorts are stable).  Queries of up
// to 16K candidates: one workgroup each (segmented sort, 1.3 ms at 6980 x 5000 against 1.9 ms
// device-wide).  A batch with a longer query: one device-wide sort on (query, score) words, whose
// cost does not depend on the query lengths (0.085 ms per million pairs; the segmented sort needs
// 0.10 / 0.15 / 0.28 ms per million at 15K / 30K / 60K candidates and 112 ms for one query of 5 M:
// tools/shapes.py).
// which form the ordering of a batch takes: one workgroup per query, in LDS (the form that can make its keys itself)?
static bool sort_in_blocks(int64_t max_len, int64_t n_queries) {
  int64_t device_wide_from = 16 << 10, block_from = 129, block_to = 8192;
  if (long long v = 0; ffx_experiment("sort_device_wide_from", &v)) device_wide_from = (int64_t)v;
  if (long long v = 0; ffx_experiment("sort_block_from", &v)) block_from = (int64_t)v;
  if (long long v = 0; ffx_experiment("sort_block_to", &v)) block_to = (int64_t)v;
  return max_len <= device_wide_from && max_len >= block_from && max_len <= std::min<int64_t>(block_to, 8192) && n_queries > 0;
}

// `made_here` (only where sort_in_blocks says so): the keys are made by the sort kernel itself (ffx::SortKeys), `keys` is unused
static int launch_segmented_sort(const float* keys, int64_t P, const int64_t* offs_dev, int64_t n_queries, DevBuf& order,
                                 DevBuf& iota, DevBuf& sorted_keys, DevBuf& sort_tmp, hipStream_t st, int64_t max_len,
                                 int32_t base = 0, const ffx::SortKeys* made_here = nullptr) {
  const size_t pb = (size_t)P;
  FFX_TRY(order.ensure(pb * sizeof(int32_t)));
  FFX_TRY(iota.ensure(pb * sizeof(int32_t)));
  int64_t device_wide_from = 16 << 10;
  if (long long v = 0; ffx_experiment("sort_device_wide_from", &v)) device_wide_from = (int64_t)v;  // A/B hook
  if (max_len > device_wide_from) {
    // words in / words out share `sorted_keys` (2 x 8 bytes per pair)
    FFX_TRY(sorted_keys.ensure(2 * pb * sizeof(unsigned long long)));
    unsigned long long* w_in = sorted_keys.as<unsigned long long>();
    unsigned long long* w_out = w_in + pb;
    ffx::sort_words_kernel<<<blocks_for(P, 256), 256, 0, st>>>(keys, offs_dev, (int)n_queries, (double)n_queries / (double)P, P, base,
                                                              w_in, iota.as<int32_t>());
    FFX_HIP(hipGetLastError());
    int q_bits = 1;
    while (q_bits < 31 && (int64_t(1) << q_bits) < n_queries) ++q_bits;
    size_t tmp = 0;
    FFX_HIP(rocprim::radix_sort_pairs(nullptr, tmp, w_in, w_out, iota.as<int32_t>(), order.as<int32_t>(), pb, 0, 32 + q_bits, st));
    FFX_TRY(sort_tmp.ensure(std::max<size_t>(tmp, 16)));
    FFX_HIP(rocprim::radix_sort_pairs(sort_tmp.p, tmp, w_in, w_out, iota.as<int32_t>(), order.as<int32_t>(), pb, 0, 32 + q_bits, st));
    return FFX_OK;
  }
  // longest query of 129 .. 8192 candidates: one workgroup per query, sorted in LDS.  Measured against
  // rocPRIM's segmented sort at 7 M pairs (tools/sort_ab.py): k = 256: 0.078 vs 0.214 ms, 1000: 0.090 vs
  // 0.150, 5000: 0.102 vs 0.365 (6980 x 5000: 0.45 vs 1.37 ms); rocPRIM wins below ~128 candidates per
  // query (k = 100: 0.115 vs 0.179, k = 10: 0.77 vs 1.63) where it packs many queries into a workgroup.
  // FFX_DEBUG_VARIANT=sort_block_from=N / sort_block_to=N move the window (A/B hook)
  int64_t block_from = 129, block_to = 8192;
  if (long long v = 0; ffx_experiment("sort_block_from", &v)) block_from = (int64_t)v;
  if (long long v = 0; ffx_experiment("sort_block_to", &v)) block_to = (int64_t)v;
  if (max_len >= block_from && max_len <= std::min<int64_t>(block_to, 8192) && n_queries > 0) {
    const unsigned nq = (unsigned)n_queries;
    ffx::SortKeys sk;
    if (made_here) sk = *made_here; else sk.keys = keys;
    if (max_len <= 256) ffx::block_sort_desc_kernel<64, 4><<<nq, 64, 0, st>>>(sk, offs_dev, base, order.as<int32_t>());
    else if (max_len <= 512) ffx::block_sort_desc_kernel<128, 4><<<nq, 128, 0, st>>>(sk, offs_dev, base, order.as<int32_t>());
    else if (max_len <= 1024) ffx::block_sort_desc_kernel<256, 4><<<nq, 256, 0, st>>>(sk, offs_dev, base, order.as<int32_t>());
    else if (max_len <= 2048) ffx::block_sort_desc_kernel<256, 8><<<nq, 256, 0, st>>>(sk, offs_dev, base, order.as<int32_t>());
    else if (max_len <= 4096) ffx::block_sort_desc_kernel<512, 8><<<nq, 512, 0, st>>>(sk, offs_dev, base, order.as<int32_t>());
    else if (max_len <= 5120) ffx::block_sort_desc_kernel<512, 10><<<nq, 512, 0, st>>>(sk, offs_dev, base, order.as<int32_t>());
    else ffx::block_sort_desc_kernel<1024, 8><<<nq, 1024, 0, st>>>(sk, offs_dev, base, order.as<int32_t>());
    FFX_HIP(hipGetLastError());
    return FFX_OK;
  }
  FFX_TRY(sorted_keys.ensure(pb * sizeof(float)));
  ffx::iota_kernel<<<blocks_for(P, 256), 256, 0, st>>>(iota.as<int32_t>(), P, base);
  FFX_HIP(hipGetLastError());
  size_t tmp = 0;
  FFX_HIP(rocprim::segmented_radix_sort_pairs_desc(nullptr, tmp, keys, sorted_keys.as<float>(), iota.as<int32_t>(),
                                                   order.as<int32_t>(), (unsigned)P, (unsigned)n_queries, offs_dev,
                                                   offs_dev + 1, 0, 32, st));
  FFX_TRY(sort_tmp.ensure(std::max<size_t>(tmp, 16)));
  FFX_HIP(rocprim::segmented_radix_sort_pairs_desc(sort_tmp.p, tmp, keys, sorted_keys.as<float>(), iota.as<int32_t>(),
                                                   order.as<int32_t>(), (unsigned)P, (unsigned)n_queries, offs_dev,
                                                   offs_dev + 1, 0, 32, st));
  return FFX_OK;
}

int ffx_plan_run(ffx_plan* pl, double alpha, uint32_t flags) {
  if (!pl) return fail(FFX_BAD_ARG, "NULL plan");
  if (!pl->have_pairs) return fail(FFX_BAD_ARG, "ffx_plan_set_pairs has not succeeded on this plan");
  if (pl->n_queries_set < pl->n_queries) return fail(FFX_BAD_ARG, "%d query vectors set, %lld queries in the pairs", pl->n_queries_set, (long long)pl->n_queries);
  ffx_index* ix = pl->idx;
  FFX_HIP(hipSetDevice(ix->device));
  const int64_t P = pl->n_pairs, R = pl->n_rows;
  hipStream_t st = pl->stream;
  // dense scores given (plan_set_given): stages 1 and 2 have happened in an earlier call; one dense score per candidate
  const bool given = pl->dense_given;
  const bool single = given || (R == P);
  const bool do_sort = (flags & FFX_FLAG_SORT) != 0;
  const bool exact = (flags & FFX_FLAG_EXACT_SUM) != 0;
  pl->ran = false;
  pl->use_derive = pl->use_derive1 = false;
  FFX_HIP(hipEventRecord(pl->ev[0], st));
  // product-quantized rows, one row per candidate, exact ADC: the three stages below are one kernel
  const bool pq_fused = !given && single && P > 0 && !ffx_debug_variant("pq_no_fuse", nullptr) && pq_exact_adc_applies(pl);
  bool pq_sorted = false;  // the fused kernel ordered the queries as well
  if (pq_fused) {
    hipEvent_t* recent_pq = pl->recent[pl->n_runs % ffx_plan::kRecent];
    if (!recent_pq[0]) {
      FFX_HIP(hipEventCreateWithFlags(&recent_pq[0], hipEventDisableSystemFence));
      FFX_HIP(hipEventCreateWithFlags(&recent_pq[1], hipEventDisableSystemFence));
    }
    FFX_HIP(hipEventRecord(pl->ev[1], st));
    FFX_HIP(hipEventRecord(recent_pq[0], st));
    FFX_TRY(launch_pq_fused(pl, (float)alpha, (float)(1.0 - alpha), flags, do_sort ? &pq_sorted : nullptr));
    FFX_HIP(hipEventRecord(pl->ev[2], st));
    FFX_HIP(hipEventRecord(recent_pq[1], st));
  }
  if (P > 0 && !pq_fused && !given) {
    // 1. candidates -> row list
    const double qpp = (double)pl->n_queries / (double)P;
    if (single && derive_kernel_exists(pl) && ix->capacity <= INT32_MAX && !pl->forced_pieces && !ffx_experiment("doc_major", nullptr) &&
        !ffx_experiment("gather", nullptr)) {
      pl->use_derive1 = true;  // candidate -> row inside gather_dot: no launch here, no row list
    } else if (single) {
      FFX_TRY(pl->desc.ensure(std::max<size_t>((size_t)P, 1) * sizeof(int2)));
      const int32_t* row_of = pl->mode == FFX_MODE_PASSAGE ? (ix->psgs_identity ? nullptr : ix->psg_row.as<int32_t>())
                                                           : (ix->docs_identity ? nullptr : ix->doc_first.as<int32_t>());
      ffx::expand_single_kernel<<<blocks_for((P + 3) / 4, 256), 256, 0, st>>>(
          pl->pair_ids.as<int32_t>(), P, pl->pair_off.as<int64_t>(), (int)pl->n_queries, qpp, row_of, ix->capacity, pl->desc.as<int2>());
    } else if (derive_applies(pl)) {
      // no row list: the scoring kernel derives its rows from the candidate lists, the row offsets (the scan of set_pairs is
      // still there) and two small tables made on the first run of a candidate list (4 B per candidate, 8 B per tile)
      pl->use_derive = true;
      if (!pl->derive_valid) {
        const int tile = derive_tile(pl);
        FFX_TRY(launch_count_scan(pl));
        FFX_TRY(pl->pair_first.ensure((size_t)P * sizeof(int32_t)));
        FFX_TRY(pl->tile_tab.ensure(((size_t)R / tile + 2) * sizeof(int2)));
        ffx::tile_index_kernel<<<blocks_for(P, 256), 256, 0, st>>>(pl->pair_ids.as<int32_t>(), P, pl->pair_off.as<int64_t>(), (int)pl->n_queries, qpp,
                                                                  ix->doc_off.as<int32_t>(), pl->seg_off.as<int32_t>(), tile, pl->pair_first.as<int32_t>(),
                                                                  pl->tile_tab.as<int2>());
        pl->derive_valid = true;
      }
    } else {
      FFX_TRY(pl->desc.ensure(std::max<size_t>((size_t)R, 1) * sizeof(int2)));  // (not made by set_pairs where rows can be derived)
      FFX_TRY(launch_count_scan(pl));
      ffx::expand_kernel<<<blocks_for(P, 256), 256, 0, st>>>(
          pl->pair_ids.as<int32_t>(), P, pl->mode, pl->pair_off.as<int64_t>(), (int)pl->n_queries, qpp,
          ix->doc_off.as<int32_t>(), ix->doc_rows.as<int32_t>(), ix->psg_row.as<int32_t>(),
          pl->seg_off.as<int32_t>(), pl->desc.as<int2>());
    }
    FFX_HIP(hipGetLastError());
  }
#ifdef FFX_EXPERIMENTS
  // FFX_DEBUG_VARIANT=doc_major: the row list is put in ROW order before gather_dot (a stable radix sort of the rows by index row:
  // candidates of different queries that name the same row stand next to each other, so the row is fetched from HBM once and
  // found in the cache by the others) and the row scores are put back afterwards.  The sort is inside the step's first stage,
  // the way back inside its third: ms_gather_dot is the kernel(s) alone.
  const bool doc_major = single && P > 0 && !pq_fused && !given && ffx_experiment("doc_major", nullptr);
  if (doc_major) {
    const size_t nb = (size_t)P * 4;
    for (DevBuf* b : {&pl->dm_keys, &pl->dm_keys2, &pl->dm_perm, &pl->dm_perm2, &pl->dm_rs}) FFX_TRY(b->ensure(nb));
    FFX_TRY(pl->dm_desc.ensure((size_t)P * sizeof(int2)));
    ffx::dm_prepare_kernel<<<blocks_for(P, 256), 256, 0, st>>>(pl->desc.as<int2>(), P, pl->dm_keys.as<uint32_t>(), pl->dm_perm.as<int32_t>());
    size_t tmp = 0;
    FFX_HIP(rocprim::radix_sort_pairs(nullptr, tmp, pl->dm_keys.as<uint32_t>(), pl->dm_keys2.as<uint32_t>(), pl->dm_perm.as<int32_t>(),
                                      pl->dm_perm2.as<int32_t>(), (size_t)P, 0, 32, st));
    FFX_TRY(pl->dm_tmp.ensure(std::max<size_t>(tmp, 16)));
    FFX_HIP(rocprim::radix_sort_pairs(pl->dm_tmp.p, tmp, pl->dm_keys.as<uint32_t>(), pl->dm_keys2.as<uint32_t>(), pl->dm_perm.as<int32_t>(),
                                      pl->dm_perm2.as<int32_t>(), (size_t)P, 0, 32, st));
    ffx::dm_gather_kernel<<<blocks_for(P, 256), 256, 0, st>>>(pl->desc.as<int2>(), pl->dm_perm2.as<int32_t>(), P, pl->dm_desc.as<int2>());
    FFX_HIP(hipGetLastError());
  }
#endif
  if (!pq_fused) FFX_HIP(hipEventRecord(pl->ev[1], st));
  hipEvent_t* recent = pl->recent[pl->n_runs % ffx_plan::kRecent];
  if (!recent[0]) {
    FFX_HIP(hipEventCreateWithFlags(&recent[0], hipEventDisableSystemFence));
    FFX_HIP(hipEventCreateWithFlags(&recent[1], hipEventDisableSystemFence));
  }
  if (!pq_fused) FFX_HIP(hipEventRecord(recent[0], st));
  // 2. gather + inner product (the hot kernel)
  float* rs = single ? pl->ff.as<float>() : pl->rs.as<float>();
#ifdef FFX_EXPERIMENTS
  if (doc_major) {
    std::swap(pl->desc, pl->dm_desc);
    const int rc_dm = launch_gather_dot(pl, exact, pl->dm_rs.as<float>());
    std::swap(pl->desc, pl->dm_desc);
    FFX_TRY(rc_dm);
    FFX_HIP(hipEventRecord(pl->ev[2], st));
    ffx::dm_scatter_kernel<<<blocks_for(P, 256), 256, 0, st>>>(pl->dm_rs.as<float>(), pl->dm_perm2.as<int32_t>(), P, rs);
    FFX_HIP(hipGetLastError());
  } else
#endif
  if (R > 0 && !pq_fused && !given) FFX_TRY(launch_gather_dot(pl, exact, rs));
#ifdef FFX_EXPERIMENTS
  if (!pq_fused && !doc_major) FFX_HIP(hipEventRecord(pl->ev[2], st));
#else
  if (!pq_fused) FFX_HIP(hipEventRecord(pl->ev[2], st));
#endif
  if (!pq_fused) FFX_HIP(hipEventRecord(recent[1], st));
  // 3. segment aggregation + interpolation.  One row per candidate, first-stage scores, queries that are ordered a workgroup
  // each: the interpolation is made by the sort kernel while it loads its keys (one launch and its gap less per run; what
  // ffx_score pays fourteen times for a batch cut into fourteen slices)
  const float a = (float)alpha;
  const float b = (float)(1.0 - alpha);
  const int64_t longest_query = (do_sort && P > 0) ? longest_segment(pl->host_pair_off.data(), pl->n_queries) : 0;
  const bool interp_in_sort = single && pl->have_sparse && do_sort && P > 0 && !pq_fused && sort_in_blocks(longest_query, pl->n_queries) &&
                              !ffx_experiment("sort_interp_apart", nullptr);
  if (P > 0 && !pq_fused) {
    if (!single) {
      ffx::reduce_kernel<<<blocks_for(P, 256), 256, 0, st>>>(
          rs, pl->seg_off.as<int32_t>(), P, pl->mode, pl->have_sparse ? pl->sparse.as<float>() : nullptr, a, b,
          pl->ff.as<float>(), pl->interp.as<float>());
    } else if (pl->have_sparse && !interp_in_sort) {
      ffx::interp_kernel<<<blocks_for((P + 3) / 4, 256), 256, 0, st>>>(pl->ff.as<float>(), pl->sparse.as<float>(), P, a, b,
                                                            pl->interp.as<float>());
    }
    FFX_HIP(hipGetLastError());
  }
  pl->last_interp = pl->have_sparse;
  FFX_HIP(hipEventRecord(pl->ev[3], st));
  // 4. per-query order: score descending, ties by position (stable radix sort)
  if (do_sort && P > 0 && !pq_sorted) {
    const float* keys = (pl->have_sparse && !(flags & FFX_FLAG_SORT_FF)) ? pl->interp.as<float>() : pl->ff.as<float>();
    ffx::SortKeys made;
    if (interp_in_sort) {
      made.dense = pl->ff.as<float>();
      made.sparse = pl->sparse.as<float>();
      made.a = a;
      made.b = b;
      made.key_is_dense = (flags & FFX_FLAG_SORT_FF) ? 1 : 0;
      made.interp_out = pl->interp.as<float>();
    }
    FFX_TRY(launch_segmented_sort(keys, P, pl->pair_off.as<int64_t>(), pl->n_queries, pl->order, pl->iota,
                                  pl->sorted_keys, pl->sort_tmp, st, longest_query, pl->order_base, interp_in_sort ? &made : nullptr));
  }
  if (pl->want_ranked && do_sort && P > 0) {
    const float* keys = (pl->have_sparse && !(flags & FFX_FLAG_SORT_FF)) ? pl->interp.as<float>() : pl->ff.as<float>();
    ffx::rank_take_kernel<<<blocks_for(P, 256), 256, 0, st>>>(keys, pl->have_carry ? pl->carry.as<int32_t>() : nullptr, pl->order.as<int32_t>(), pl->order_base, P,
                                                           pl->ranked.as<float>(), pl->carry_ranked.as<int32_t>());
    FFX_HIP(hipGetLastError());
  }
  FFX_HIP(hipEventRecord(pl->ev[4], st));
  pl->ran = true;
  pl->ran_sort = do_sort;
  ++pl->n_runs;
  if (given) {  // one shot: the plan holds no row list a second run could start from
    pl->dense_given = false;
    pl->have_pairs = false;
  }
  return FFX_OK;
}

// gather_dot's duration (HIP events on the plan's stream) of the last n runs of ffx_plan_run, oldest first, for a caller
// that queued them without waiting in between; waits for the stream.  n <= 256 and <= the runs so far.
int ffx_plan_recent_gather_ms(ffx_plan* pl, int32_t n, float* out_ms) {
  if (!pl || n < 0 || (n > 0 && !out_ms)) return fail(FFX_BAD_ARG, "bad argument");
  if (n > ffx_plan::kRecent || (uint64_t)n > pl->n_runs) return fail(FFX_BAD_ARG, "only the last %d runs are kept (%llu so far)", ffx_plan::kRecent, (unsigned long long)pl->n_runs);
  FFX_HIP(hipSetDevice(pl->device));
  FFX_HIP(hipStreamSynchronize(pl->stream));
  for (int32_t i = 0; i < n; ++i) {
    hipEvent_t* pair = pl->recent[(pl->n_runs - (uint64_t)n + (uint64_t)i) % ffx_plan::kRecent];
    FFX_HIP(hipEventElapsedTime(&out_ms[i], pair[0], pair[1]));
  }
  return FFX_OK;
}

// Index._early_stopping (index/base.py:316-387), device-resident: the per-query state (still active?
// dense and interpolated score of every candidate scored so far) never leaves HBM between depth
// intervals; per interval the host reads back two counters (how many candidates / rows the interval
// has: they size the launches) and nothing else.
int ffx_plan_early_stopping(ffx_plan* pl, int32_t cutoff, double alpha, const int32_t* depths, int32_t n_depths, uint32_t flags,
                            int64_t* bad_pair, int64_t* n_scored_pairs, int32_t* n_intervals) {
  if (bad_pair) *bad_pair = -1;
  if (n_scored_pairs) *n_scored_pairs = 0;
  if (n_intervals) *n_intervals = 0;
  if (!pl || (n_depths > 0 && !depths) || n_depths < 0) return fail(FFX_BAD_ARG, "bad argument");
  if (!pl->lazy_pairs && !pl->have_pairs) return fail(FFX_BAD_ARG, "set the candidate lists first (ffx_plan_set_pairs_lazy)");
  if (!pl->have_sparse) return fail(FFX_BAD_ARG, "early stopping needs first-stage scores");
  if (cutoff < 1) return fail(FFX_BAD_ARG, "cutoff must be positive");
  if (pl->n_queries_set < pl->n_queries) return fail(FFX_BAD_ARG, "%d query vectors set, %lld queries in the pairs", pl->n_queries_set, (long long)pl->n_queries);
  ffx_index* ix = pl->idx;
  FFX_HIP(hipSetDevice(pl->device));
  hipStream_t st = pl->stream;
  const int64_t Q = pl->n_queries, P = pl->n_pairs;
  std::vector<int32_t> ds(depths, depths + n_depths);
  std::sort(ds.begin(), ds.end());
  pl->ran = false;
  if (!pl->es_child) {
    FFX_TRY(ffx_plan_create(ix, &pl->es_child));
    pl->es_child->stream = st;  // everything of one early-stopping call is ordered on this plan's stream
  }
  ffx_plan* ch = pl->es_child;
  ch->stream = st;
  const size_t pb = std::max<size_t>((size_t)P, 1), qb = (size_t)Q + 1;
  FFX_TRY(pl->es_active.ensure(qb * sizeof(int32_t)));
  FFX_TRY(pl->es_cnt.ensure(qb * sizeof(int64_t)));
  FFX_TRY(pl->es_pos.ensure(pb * sizeof(int32_t)));
  FFX_TRY(ch->pair_off.ensure(qb * sizeof(int64_t)));
  FFX_TRY(ch->pair_ids.ensure(pb * sizeof(int32_t)));
  FFX_TRY(ch->sparse.ensure(pb * sizeof(float)));
  FFX_TRY(ch->cnt.ensure((pb + 1) * sizeof(int32_t)));
  FFX_TRY(ch->seg_off.ensure((pb + 1) * sizeof(int32_t)));
  FFX_TRY(ch->ff.ensure(pb * sizeof(float)));
  FFX_TRY(ch->interp.ensure(pb * sizeof(float)));
  const size_t q_bytes = (size_t)pl->n_queries_set * ix->qdim * sizeof(float);
  FFX_TRY(ch->queries.ensure(std::max<size_t>(q_bytes, 16)));
  if (q_bytes) FFX_HIP(hipMemcpyAsync(ch->queries.p, pl->queries.p, q_bytes, hipMemcpyDeviceToDevice, st));
  ch->n_queries_set = pl->n_queries_set;
  FFX_HIP(hipEventRecord(pl->ev[0], st));
  FFX_HIP(hipEventRecord(pl->ev[1], st));
  if (P > 0) {
    ffx::fill_u32_kernel<<<blocks_for(P, 256), 256, 0, st>>>(pl->ff.as<uint32_t>(), P, 0x7FC00000u);  // NaN: not scored
    ffx::fill_u32_kernel<<<blocks_for(P, 256), 256, 0, st>>>(pl->interp.as<uint32_t>(), P, 0x7FC00000u);
  }
  if (Q > 0) ffx::fill_u32_kernel<<<blocks_for(Q, 256), 256, 0, st>>>(pl->es_active.as<uint32_t>(), Q, 1u);
  FFX_HIP(hipGetLastError());
  const float fa = (float)alpha, fb = (float)(1.0 - alpha);
  int32_t a = 0, intervals = 0;
  int64_t scored = 0, rows_total = 0;
  for (int32_t b : ds) {
    if (b < cutoff || Q == 0 || P == 0) continue;
    if (b <= a) break;  // a repeated depth: the reference asks for ranks [a, a), gets no rows and leaves the loop (base.py:362-369)
    if (a > 0) {
      ffx::es_criterion_kernel<<<(unsigned)Q, 256, 0, st>>>(pl->interp.as<float>(), pl->ff.as<float>(), pl->sparse.as<float>(),
                                                           pl->pair_off.as<int64_t>(), a, cutoff, fa, fb, alpha, 1.0 - alpha,
                                                           (flags & FFX_FLAG_ES_BOUND_F32) ? 0 : 1, pl->es_active.as<int32_t>());
      FFX_HIP(hipGetLastError());
    }
    ffx::es_count_kernel<<<blocks_for(Q + 1, 256), 256, 0, st>>>(pl->pair_off.as<int64_t>(), (int)Q, pl->es_active.as<int32_t>(), a, b,
                                                                pl->es_cnt.as<int64_t>());
    FFX_HIP(hipGetLastError());
    size_t tmp = 0;
    FFX_HIP(rocprim::exclusive_scan(nullptr, tmp, pl->es_cnt.as<int64_t>(), ch->pair_off.as<int64_t>(), (int64_t)0, (size_t)(Q + 1),
                                    rocprim::plus<int64_t>(), st));
    FFX_TRY(pl->es_scan_tmp.ensure(std::max<size_t>(tmp, 16)));
    FFX_HIP(rocprim::exclusive_scan(pl->es_scan_tmp.p, tmp, pl->es_cnt.as<int64_t>(), ch->pair_off.as<int64_t>(), (int64_t)0, (size_t)(Q + 1),
                                    rocprim::plus<int64_t>(), st));
    int64_t n_sub = 0;  // counter 1 of 2: candidates of this interval
    FFX_TRY(ffx_d2h(pl->device, &n_sub, ch->pair_off.as<int64_t>() + Q, sizeof n_sub, st));
    FFX_HIP(hipStreamSynchronize(st));
    if (n_sub == 0) break;  // no pairs are left (base.py:368-369)
    ffx::es_fill_kernel<<<(unsigned)Q, 256, 0, st>>>(pl->pair_off.as<int64_t>(), pl->pair_ids.as<int32_t>(), pl->sparse.as<float>(),
                                                    ch->pair_off.as<int64_t>(), a, ch->pair_ids.as<int32_t>(), ch->sparse.as<float>(),
                                                    pl->es_pos.as<int32_t>());
    FFX_HIP(hipGetLastError());
    ch->n_queries = Q;
    ch->n_pairs = n_sub;
    ch->mode = pl->mode;
    ch->have_sparse = true;
    ch->have_pairs = false;
    ch->q_hint_valid = false;  // (the interval's lists are made on the device: the kernel falls back to its own search)
    // get_indices for the interval only: an unknown id is an IndexError when (and only when) it is reached
    FFX_HIP(hipMemsetAsync(ch->flags.p, 0xFF, sizeof(unsigned long long), st));
    FFX_TRY(launch_count_scan(ch));
    unsigned long long bad = ~0ull;
    int32_t rows = 0;  // counter 2 of 2: index rows the interval gathers
    FFX_TRY(ffx_d2h(pl->device, &bad, ch->flags.p, sizeof bad, st));
    FFX_TRY(ffx_d2h(pl->device, &rows, ch->seg_off.as<int32_t>() + n_sub, sizeof rows, st));
    FFX_HIP(hipStreamSynchronize(st));
    if (bad != ~0ull) {
      int32_t at = -1;
      FFX_TRY(ffx_d2h_now(pl->device, &at, pl->es_pos.as<int32_t>() + bad, sizeof at));
      if (bad_pair) *bad_pair = at;
      return fail(FFX_NOT_FOUND, "pair %d names an ordinal with no rows in the index", at);
    }
    if (rows < 0) return fail(FFX_BAD_ARG, "more than 2^31-1 gathered rows in one interval: split the queries");
    ch->n_rows = rows;
    const size_t rb = std::max<size_t>((size_t)rows, 1);
    FFX_TRY(ch->desc.ensure(rb * sizeof(int2)));
    if (ch->n_rows != ch->n_pairs) FFX_TRY(ch->rs.ensure(rb * sizeof(float)));
    ch->have_pairs = true;
    ch->derive_valid = false;
    FFX_TRY(ffx_plan_run(ch, alpha, flags & FFX_FLAG_EXACT_SUM));
    ffx::es_scatter_kernel<<<blocks_for(n_sub, 256), 256, 0, st>>>(ch->ff.as<float>(), ch->interp.as<float>(), pl->es_pos.as<int32_t>(), n_sub,
                                                                  pl->ff.as<float>(), pl->interp.as<float>());
    FFX_HIP(hipGetLastError());
    pl->variant = ch->variant;
    scored += n_sub;
    rows_total += rows;
    a = b;
    ++intervals;
  }
  for (int i = 2; i < 5; ++i) FFX_HIP(hipEventRecord(pl->ev[i], st));
  pl->n_rows = rows_total;
  pl->ran = true;
  pl->ran_sort = false;
  pl->last_interp = true;
  if (n_scored_pairs) *n_scored_pairs = scored;
  if (n_intervals) *n_intervals = intervals;
  return FFX_OK;
}

int ffx_plan_sync(ffx_plan* pl) {
  if (!pl) return fail(FFX_BAD_ARG, "NULL plan");
  FFX_HIP(hipSetDevice(pl->idx->device));
  FFX_HIP(hipStreamSynchronize(pl->stream));
  return FFX_OK;
}

int ffx_plan_fetch(ffx_plan* pl, float* out_ff, float* out_interp, int32_t* out_order) {
  if (!pl) return fail(FFX_BAD_ARG, "NULL plan");
  if (!pl->ran) return fail(FFX_BAD_ARG, "ffx_plan_run has not completed on this plan");
  FFX_HIP(hipSetDevice(pl->idx->device));
  const size_t P = (size_t)pl->n_pairs;
  if (out_interp && !pl->last_interp) return fail(FFX_BAD_ARG, "no sparse scores were set: nothing to interpolate");
  if (out_order && !pl->ran_sort) return fail(FFX_BAD_ARG, "the last run did not request FFX_FLAG_SORT");
  if (P > 0) {
    pl->hstage_used = 0;
    const size_t each = P * sizeof(float), slot = (each + 63) & ~size_t(63);
    const size_t wanted = (out_ff ? 1 : 0) + (out_interp ? 1 : 0) + (out_order ? 1 : 0);
    char* back = wanted * slot <= ffx_plan::kStageBytes / 2 ? pl->stage(3 * slot <= ffx_plan::kStageBytes ? 3 * slot : wanted * slot) : nullptr;
    if (back) {  // small result: up to three copies into page-locked memory, one wait, then hand the bytes over
      const bool all_slots = 3 * slot <= ffx_plan::kStageBytes;
      char* b_ff = back;
      char* b_in = all_slots ? back + slot : back + (out_ff ? slot : 0);
      char* b_or = all_slots ? back + 2 * slot : b_in + (out_interp ? slot : 0);
      if (out_ff) FFX_HIP(hipMemcpyAsync(b_ff, pl->ff.p, each, hipMemcpyDeviceToHost, pl->stream));
      if (out_interp) FFX_HIP(hipMemcpyAsync(b_in, pl->interp.p, each, hipMemcpyDeviceToHost, pl->stream));
      if (out_order) FFX_HIP(hipMemcpyAsync(b_or, pl->order.p, each, hipMemcpyDeviceToHost, pl->stream));
      FFX_HIP(hipStreamSynchronize(pl->stream));
      if (out_ff) memcpy(out_ff, b_ff, each);
      if (out_interp) memcpy(out_interp, b_in, each);
      if (out_order) memcpy(out_order, b_or, each);
      return FFX_OK;
    }
    if (out_ff) FFX_TRY(ffx_d2h(pl->device, out_ff, pl->ff.p, P * sizeof(float), pl->stream));
    if (out_interp) FFX_TRY(ffx_d2h(pl->device, out_interp, pl->interp.p, P * sizeof(float), pl->stream));
    if (out_order) FFX_TRY(ffx_d2h(pl->device, out_order, pl->order.p, P * sizeof(int32_t), pl->stream));
  }
  FFX_HIP(hipStreamSynchronize(pl->stream));
  return FFX_OK;
}

// After ffx_plan_early_stopping: only the candidates that were scored, compacted on the device -- their positions in
// the candidate list (ascending, so still grouped by query) and their dense scores.  The dense array holds NaN for
// every candidate that was not reached; at depth 5000 and a stop after the first interval that is 98 % of it, and
// fetching and scanning it on the host costs more than the scoring.  *n_scored = how many there are; up to `cap` are
// written (call with cap = the count ffx_plan_early_stopping reported).
int ffx_plan_fetch_scored(ffx_plan* pl, int64_t cap, int32_t* out_pos, float* out_ff, int64_t* n_scored) {
  if (!pl || !n_scored || cap < 0 || (cap > 0 && (!out_pos || !out_ff))) return fail(FFX_BAD_ARG, "bad argument");
  if (!pl->ran) return fail(FFX_BAD_ARG, "nothing has run on this plan");
  FFX_HIP(hipSetDevice(pl->device));
  hipStream_t st = pl->stream;
  const int64_t P = pl->n_pairs;
  *n_scored = 0;
  if (P == 0) return FFX_OK;
  const size_t pb = (size_t)P;
  FFX_TRY(pl->es_pos.ensure(pb * sizeof(int32_t)));
  FFX_TRY(pl->iota.ensure(pb * sizeof(float) + 64));
  FFX_TRY(pl->es_cnt.ensure(64));
  auto scored = rocprim::make_transform_iterator(pl->ff.as<float>(), [] __device__(float v) { return v == v; });
  rocprim::counting_iterator<int32_t> positions(0);
  unsigned long long* d_count = pl->es_cnt.as<unsigned long long>();
  size_t tmp = 0;
  FFX_HIP(rocprim::select(nullptr, tmp, positions, scored, pl->es_pos.as<int32_t>(), d_count, pb, st));
  FFX_TRY(pl->es_scan_tmp.ensure(std::max<size_t>(tmp, 16)));
  FFX_HIP(rocprim::select(pl->es_scan_tmp.p, tmp, positions, scored, pl->es_pos.as<int32_t>(), d_count, pb, st));
  unsigned long long count = 0;
  FFX_TRY(ffx_d2h(pl->device, &count, d_count, sizeof count, st));
  FFX_HIP(hipStreamSynchronize(st));
  *n_scored = (int64_t)count;
  const int64_t n = std::min<int64_t>((int64_t)count, cap);
  if (n == 0) return FFX_OK;
  float* packed = pl->iota.as<float>();
  ffx::take_f32_kernel<<<blocks_for(n, 256), 256, 0, st>>>(pl->ff.as<float>(), pl->es_pos.as<int32_t>(), n, packed);
  FFX_HIP(hipGetLastError());
  FFX_TRY(ffx_d2h(pl->device, out_pos, pl->es_pos.p, (size_t)n * sizeof(int32_t), st));
  FFX_TRY(ffx_d2h(pl->device, out_ff, packed, (size_t)n * sizeof(float), st));
  FFX_HIP(hipStreamSynchronize(st));
  return FFX_OK;
}

int ffx_plan_stats(ffx_plan* pl, ffx_stats* out) {
  if (!pl || !out) return fail(FFX_BAD_ARG, "NULL argument");
  if (!pl->ran) return fail(FFX_BAD_ARG, "ffx_plan_run has not completed on this plan");
  FFX_HIP(hipSetDevice(pl->idx->device));
  FFX_HIP(hipStreamSynchronize(pl->stream));
  memset(out, 0, sizeof *out);
  out->n_queries = pl->n_queries;
  out->n_pairs = pl->n_pairs;
  out->n_rows = pl->n_rows;
  const double d = pl->idx->qdim;
  out->algorithmic_bytes = (double)pl->n_rows * ((double)pl->idx->dim * (double)pl->idx->elem + 4) + (double)pl->n_pairs * 4 * (1 + (pl->last_interp ? 2 : 0)) +
                           (double)pl->n_queries * d * 4 + (double)(pl->n_queries + 1) * 8;
  FFX_HIP(hipEventElapsedTime(&out->ms_expand, pl->ev[0], pl->ev[1]));
  FFX_HIP(hipEventElapsedTime(&out->ms_gather_dot, pl->ev[1], pl->ev[2]));
  FFX_HIP(hipEventElapsedTime(&out->ms_reduce, pl->ev[2], pl->ev[3]));
  FFX_HIP(hipEventElapsedTime(&out->ms_sort, pl->ev[3], pl->ev[4]));
  FFX_HIP(hipEventElapsedTime(&out->ms_total, pl->ev[0], pl->ev[4]));
  out->kernel_variant = pl->variant;
  return FFX_OK;
}

int ffx_plan_device_ptr(ffx_plan* pl, int which, void** ptr, int64_t* n_elems) {
  if (!pl || !ptr) return fail(FFX_BAD_ARG, "NULL argument");
  const DevBuf* b = nullptr;
  int64_t n = pl->n_pairs;
  switch (which) {
    case FFX_BUF_FF: b = &pl->ff; break;
    case FFX_BUF_INTERP: b = &pl->interp; break;
    case FFX_BUF_ORDER: b = &pl->order; break;
    case FFX_BUF_QUERIES: b = &pl->queries; n = (int64_t)pl->n_queries_set * pl->idx->qdim; break;
    case FFX_BUF_PAIR_IDS: b = &pl->pair_ids; break;
    case FFX_BUF_SPARSE: b = &pl->sparse; break;
    default: return fail(FFX_BAD_ARG, "unknown buffer selector %d", which);
  }
  *ptr = b->p;
  if (n_elems) *n_elems = b->p ? n : 0;
  return FFX_OK;
}

int ffx_plan_stream(ffx_plan* pl, void** stream) {
  if (!pl || !stream) return fail(FFX_BAD_ARG, "NULL argument");
  *stream = (void*)pl->stream;
  return FFX_OK;
}

// ---- merge -----------------------------------------------------------------------------------------
}  // extern "C"

struct ffx_merge {
  int device = 0;
  hipStream_t own_stream = nullptr, last_stream = nullptr;
  DevBuf pair_off, slot, sparse, ff, interp, order, iota, sorted_keys, sort_tmp;
  int64_t n_queries = 0, n_pairs = 0, max_len = 0;
  int32_t order_base = 0;  // added to the positions of the per-query order (the owned block as a slice of a larger list)
  bool have_sparse = false, ready = false, ran = false, ran_sort = false;
};

extern "C" {

int ffx_merge_create(int device, ffx_merge** out) {
  if (!out) return fail(FFX_BAD_ARG, "out is NULL");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    (void)hipGetLastError();
    return fail(FFX_NO_DEVICE, "no HIP device: libffx has no CPU fallback");
  }
  if (device < 0 || device >= n) return fail(FFX_BAD_ARG, "device %d out of range (%d devices)", device, n);
  FFX_HIP(hipSetDevice(device));
  ffx_merge* m = new (std::nothrow) ffx_merge();
  if (!m) return fail(FFX_OOM, "host allocation failed");
  m->device = device;
  if (hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete m;
    return fail(FFX_HIP_ERROR, "stream creation failed");
  }
  m->last_stream = m->own_stream;
  *out = m;
  return FFX_OK;
}

int ffx_merge_destroy(ffx_merge* m) {
  if (!m) return FFX_OK;
  (void)hipSetDevice(m->device);
  if (m->last_stream) (void)hipStreamSynchronize(m->last_stream);
  for (DevBuf* b : {&m->pair_off, &m->slot, &m->sparse, &m->ff, &m->interp, &m->order, &m->iota, &m->sorted_keys, &m->sort_tmp})
    b->release();
  if (m->own_stream) (void)hipStreamDestroy(m->own_stream);
  delete m;
  return FFX_OK;
}

int ffx_merge_set(ffx_merge* m, const int64_t* pair_off, int64_t n_queries, const int32_t* recv_slot, const float* sparse) {
  if (!m || !pair_off || n_queries < 0) return fail(FFX_BAD_ARG, "bad argument");
  if (pair_off[0] != 0) return fail(FFX_BAD_ARG, "pair_off[0] must be 0");
  for (int64_t q = 0; q < n_queries; ++q)
    if (pair_off[q + 1] < pair_off[q]) return fail(FFX_BAD_ARG, "pair_off not monotone at query %lld", (long long)q);
  const int64_t P = pair_off[n_queries];
  if (P > INT32_MAX - 1) return fail(FFX_BAD_ARG, "too many pairs");
  if (P > 0 && !recv_slot) return fail(FFX_BAD_ARG, "recv_slot is NULL");
  for (int64_t p = 0; p < P; ++p)
    if (recv_slot[p] < 0 || recv_slot[p] >= P) return fail(FFX_BAD_ARG, "recv_slot[%lld] out of range", (long long)p);
  FFX_HIP(hipSetDevice(m->device));
  m->ready = m->ran = false;
  const size_t pb = std::max<size_t>((size_t)P, 1);
  FFX_TRY(m->pair_off.ensure((size_t)(n_queries + 1) * sizeof(int64_t)));
  FFX_TRY(m->slot.ensure(pb * sizeof(int32_t)));
  FFX_TRY(m->ff.ensure(pb * sizeof(float)));
  FFX_TRY(ffx_h2d_now(m->device, m->pair_off.p, pair_off, (size_t)(n_queries + 1) * sizeof(int64_t)));
  if (P > 0) FFX_TRY(ffx_h2d_now(m->device, m->slot.p, recv_slot, (size_t)P * sizeof(int32_t)));
  m->have_sparse = sparse != nullptr;
  if (sparse) {
    FFX_TRY(m->sparse.ensure(pb * sizeof(float)));
    FFX_TRY(m->interp.ensure(pb * sizeof(float)));
    if (P > 0) FFX_TRY(ffx_h2d_now(m->device, m->sparse.p, sparse, (size_t)P * sizeof(float)));
  }
  m->n_queries = n_queries;
  m->n_pairs = P;
  m->max_len = longest_segment(pair_off, n_queries);
  m->order_base = 0;
  m->ready = true;
  return FFX_OK;
}

// ffx_merge_set with the arrival slots already in device memory (the sharded scorer routes on the device); the
// offsets and the first-stage scores of the owned queries come from the host as before.  The slots are checked on
// the device (each inside 0 .. P-1).
int ffx_merge_set_device(ffx_merge* m, const int64_t* pair_off, int64_t n_queries, const int32_t* dev_recv_slot, const float* sparse) {
  if (!m || !pair_off || n_queries < 0) return fail(FFX_BAD_ARG, "bad argument");
  if (pair_off[0] != 0) return fail(FFX_BAD_ARG, "pair_off[0] must be 0");
  for (int64_t q = 0; q < n_queries; ++q)
    if (pair_off[q + 1] < pair_off[q]) return fail(FFX_BAD_ARG, "pair_off not monotone at query %lld", (long long)q);
  const int64_t P = pair_off[n_queries];
  if (P > INT32_MAX - 1) return fail(FFX_BAD_ARG, "too many pairs");
  if (P > 0 && !dev_recv_slot) return fail(FFX_BAD_ARG, "recv_slot is NULL");
  FFX_HIP(hipSetDevice(m->device));
  m->ready = m->ran = false;
  const size_t pb = std::max<size_t>((size_t)P, 1);
  FFX_TRY(m->pair_off.ensure((size_t)(n_queries + 1) * sizeof(int64_t)));
  FFX_TRY(m->slot.ensure(pb * sizeof(int32_t)));
  FFX_TRY(m->ff.ensure(pb * sizeof(float)));
  FFX_TRY(m->sort_tmp.ensure(64));
  hipStream_t st = m->own_stream;
  FFX_TRY(ffx_h2d(m->device, m->pair_off.p, pair_off, (size_t)(n_queries + 1) * sizeof(int64_t), st));
  unsigned long long bad = ~0ull;
  if (P > 0) {
    FFX_HIP(hipMemcpyAsync(m->slot.p, dev_recv_slot, (size_t)P * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    unsigned long long* flag = m->sort_tmp.as<unsigned long long>();
    FFX_HIP(hipMemsetAsync(flag, 0xFF, sizeof *flag, st));
    ffx::check_range_kernel<<<blocks_for(P, 256), 256, 0, st>>>(m->slot.as<int32_t>(), P, (int32_t)P, flag);
    FFX_HIP(hipGetLastError());
    FFX_TRY(ffx_d2h(m->device, &bad, flag, sizeof bad, st));
  }
  m->have_sparse = sparse != nullptr;
  if (sparse) {
    FFX_TRY(m->sparse.ensure(pb * sizeof(float)));
    FFX_TRY(m->interp.ensure(pb * sizeof(float)));
    if (P > 0) FFX_TRY(ffx_h2d(m->device, m->sparse.p, sparse, (size_t)P * sizeof(float), st));
  }
  FFX_HIP(hipStreamSynchronize(st));
  if (bad != ~0ull) return fail(FFX_BAD_ARG, "recv_slot[%llu] out of range", bad);
  m->n_queries = n_queries;
  m->n_pairs = P;
  m->max_len = longest_segment(pair_off, n_queries);
  m->order_base = 0;
  m->ready = true;
  return FFX_OK;
}

int ffx_merge_run(ffx_merge* m, const float* recv_dev, double alpha, uint32_t flags, void* stream) {
  if (!m) return fail(FFX_BAD_ARG, "NULL merge");
  if (!m->ready) return fail(FFX_BAD_ARG, "ffx_merge_set has not succeeded on this handle");
  if (m->n_pairs > 0 && !recv_dev) return fail(FFX_BAD_ARG, "recv_dev is NULL");
  FFX_HIP(hipSetDevice(m->device));
  hipStream_t st = stream ? (hipStream_t)stream : m->own_stream;
  m->last_stream = st;
  const int64_t P = m->n_pairs;
  const bool do_sort = (flags & FFX_FLAG_SORT) != 0;
  if (P > 0) {
    // queries that are ordered a workgroup each: the scatter out of the receive buffer and the interpolation are made by the
    // sort kernel while it loads its keys -- the owner-side half of a step is ONE launch
    const bool in_sort = do_sort && sort_in_blocks(m->max_len, m->n_queries) && !ffx_experiment("sort_interp_apart", nullptr);
    if (!in_sort) {
      ffx::merge_kernel<<<blocks_for(P, 256), 256, 0, st>>>(recv_dev, m->slot.as<int32_t>(),
                                                           m->have_sparse ? m->sparse.as<float>() : nullptr, P, (float)alpha,
                                                           (float)(1.0 - alpha), m->ff.as<float>(), m->interp.as<float>());
      FFX_HIP(hipGetLastError());
    }
    if (do_sort) {
      const float* keys = (m->have_sparse && !(flags & FFX_FLAG_SORT_FF)) ? m->interp.as<float>() : m->ff.as<float>();
      ffx::SortKeys made;
      if (in_sort) {
        made.dense = recv_dev;
        made.slot = m->slot.as<int32_t>();
        made.sparse = m->have_sparse ? m->sparse.as<float>() : nullptr;
        made.a = (float)alpha;
        made.b = (float)(1.0 - alpha);
        made.key_is_dense = (flags & FFX_FLAG_SORT_FF) ? 1 : 0;
        made.ff_out = m->ff.as<float>();
        made.interp_out = m->interp.as<float>();
      }
      FFX_TRY(launch_segmented_sort(keys, P, m->pair_off.as<int64_t>(), m->n_queries, m->order, m->iota, m->sorted_keys,
                                    m->sort_tmp, st, m->max_len, m->order_base, in_sort ? &made : nullptr));
    }
  }
  m->ran = true;
  m->ran_sort = do_sort;
  return FFX_OK;
}

int ffx_merge_sync(ffx_merge* m) {
  if (!m) return fail(FFX_BAD_ARG, "NULL merge");
  FFX_HIP(hipSetDevice(m->device));
  FFX_HIP(hipStreamSynchronize(m->last_stream));
  return FFX_OK;
}

int ffx_merge_fetch(ffx_merge* m, float* out_ff, float* out_interp, int32_t* out_order) {
  if (!m) return fail(FFX_BAD_ARG, "NULL merge");
  if (!m->ran) return fail(FFX_BAD_ARG, "ffx_merge_run has not completed on this handle");
  if (out_interp && !m->have_sparse) return fail(FFX_BAD_ARG, "no sparse scores were set: nothing to interpolate");
  if (out_order && !m->ran_sort) return fail(FFX_BAD_ARG, "the last run did not request FFX_FLAG_SORT");
  FFX_HIP(hipSetDevice(m->device));
  FFX_HIP(hipStreamSynchronize(m->last_stream));
  const size_t P = (size_t)m->n_pairs;
  if (P > 0) {
    if (out_ff) FFX_TRY(ffx_d2h_now(m->device, out_ff, m->ff.p, P * sizeof(float)));
    if (out_interp) FFX_TRY(ffx_d2h_now(m->device, out_interp, m->interp.p, P * sizeof(float)));
    if (out_order) FFX_TRY(ffx_d2h_now(m->device, out_order, m->order.p, P * sizeof(int32_t)));
  }
  return FFX_OK;
}

int ffx_merge_device_ptr(ffx_merge* m, int which, void** ptr, int64_t* n_elems) {
  if (!m || !ptr) return fail(FFX_BAD_ARG, "NULL argument");
  const DevBuf* b = nullptr;
  switch (which) {
    case FFX_BUF_FF: b = &m->ff; break;
    case FFX_BUF_INTERP: b = &m->interp; break;
    case FFX_BUF_ORDER: b = &m->order; break;
    case FFX_BUF_SPARSE: b = &m->sparse; break;
    default: return fail(FFX_BAD_ARG, "unknown buffer selector %d", which);
  }
  *ptr = b->p;
  if (n_elems) *n_elems = b->p ? m->n_pairs : 0;
  return FFX_OK;
}

}  // extern "C"

// ---- ffx_internal.h: the owner-side state written in place by the sharded scorer (comm.hip) -------------------------
int ffx_merge_create_on(int device, ffx_merge** out) {
  if (!out) return fail(FFX_BAD_ARG, "out is NULL");
  *out = nullptr;
  FFX_HIP(hipSetDevice(device));
  ffx_merge* m = new (std::nothrow) ffx_merge();
  if (!m) return fail(FFX_OOM, "host allocation failed");
  m->device = device;
  *out = m;
  return FFX_OK;
}

int ffx_merge_adopt_begin(ffx_merge* m, int64_t n_queries, int64_t n_pairs, bool have_sparse, int64_t** d_pair_off, int32_t** d_slot,
                          float** d_sparse) {
  if (!m || n_queries < 0 || n_pairs < 0 || !d_pair_off || !d_slot || !d_sparse) return fail(FFX_BAD_ARG, "bad argument");
  if (n_pairs > INT32_MAX - 1) return fail(FFX_BAD_ARG, "too many pairs");
  FFX_HIP(hipSetDevice(m->device));
  m->ready = m->ran = false;
  const size_t pb = std::max<size_t>((size_t)n_pairs, 1);
  FFX_TRY(m->pair_off.ensure((size_t)(n_queries + 1) * sizeof(int64_t)));
  FFX_TRY(m->slot.ensure(pb * sizeof(int32_t)));
  FFX_TRY(m->ff.ensure(pb * sizeof(float)));
  m->have_sparse = have_sparse;
  if (have_sparse) {
    FFX_TRY(m->sparse.ensure(pb * sizeof(float)));
    FFX_TRY(m->interp.ensure(pb * sizeof(float)));
  }
  m->n_queries = n_queries;
  m->n_pairs = n_pairs;
  *d_pair_off = m->pair_off.as<int64_t>();
  *d_slot = m->slot.as<int32_t>();
  *d_sparse = have_sparse ? m->sparse.as<float>() : nullptr;
  return FFX_OK;
}

int ffx_merge_adopt_finish(ffx_merge* m, int64_t longest_query, int32_t order_base) {
  if (!m || longest_query < 0) return fail(FFX_BAD_ARG, "bad argument");
  m->max_len = longest_query;
  m->order_base = order_base;
  m->ready = true;
  return FFX_OK;
}

int ffx_merge_fetch_async(ffx_merge* m, float* out_ff, float* out_interp, int32_t* out_order, hipStream_t stream) {
  if (!m) return fail(FFX_BAD_ARG, "NULL merge");
  if (!m->ran) return fail(FFX_BAD_ARG, "ffx_merge_run has not been issued on this handle");
  if (out_interp && !m->have_sparse) return fail(FFX_BAD_ARG, "no sparse scores were set: nothing to interpolate");
  if (out_order && !m->ran_sort) return fail(FFX_BAD_ARG, "the last run did not request FFX_FLAG_SORT");
  FFX_HIP(hipSetDevice(m->device));
  const size_t P = (size_t)m->n_pairs;
  if (P > 0) {
    if (out_ff) FFX_TRY(ffx_d2h(m->device, out_ff, m->ff.p, P * sizeof(float), stream));
    if (out_interp) FFX_TRY(ffx_d2h(m->device, out_interp, m->interp.p, P * sizeof(float), stream));
    if (out_order) FFX_TRY(ffx_d2h(m->device, out_order, m->order.p, P * sizeof(int32_t), stream));
  }
  return FFX_OK;
}

int64_t ffx_merge_pairs(const ffx_merge* m) { return m ? m->n_pairs : 0; }

extern "C" {

// ---- stand-alone epilogue ---------------------------------------------------------------------------
int ffx_aggregate(int device, const float* row_scores, const int64_t* seg_off, int64_t n_pairs, int mode,
                  const float* sparse, double alpha, const int64_t* pair_off, int64_t n_queries, uint32_t flags,
                  float* out_ff, float* out_interp, int32_t* out_order) {
  if (n_pairs < 0 || (n_pairs > 0 && (!row_scores || !out_ff))) return fail(FFX_BAD_ARG, "bad argument");
  if (mode < FFX_MODE_PASSAGE || mode > FFX_MODE_AVEP) return fail(FFX_BAD_ARG, "unknown mode %d", mode);
  if (out_interp && !sparse) return fail(FFX_BAD_ARG, "no sparse scores: nothing to interpolate");
  const bool do_sort = (flags & FFX_FLAG_SORT) != 0;
  if (do_sort && (!pair_off || !out_order || n_queries < 0)) return fail(FFX_BAD_ARG, "sorting needs pair_off and out_order");
  if (n_pairs == 0) return FFX_OK;
  if (n_pairs > INT32_MAX - 1) return fail(FFX_BAD_ARG, "bad segment offsets");
  // seg_off == NULL: one row per candidate (the row scores are the candidates' scores)
  const int64_t R = seg_off ? seg_off[n_pairs] : n_pairs;
  std::vector<int32_t> off32;
  if (seg_off) {
    if (seg_off[0] != 0 || R > INT32_MAX) return fail(FFX_BAD_ARG, "bad segment offsets");
    off32.resize((size_t)n_pairs + 1);
    for (int64_t p = 0; p <= n_pairs; ++p) {
      if (p > 0 && seg_off[p] <= seg_off[p - 1]) return fail(FFX_BAD_ARG, "candidate %lld has no rows", (long long)(p - 1));
      off32[(size_t)p] = (int32_t)seg_off[p];
    }
  }
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
    (void)hipGetLastError();
    return fail(FFX_NO_DEVICE, "no HIP device: libffx has no CPU fallback");
  }
  FFX_HIP(hipSetDevice(device));
  // the inputs and results of the call share ONE device block (an allocation and a release per buffer cost more
  // than the kernels of a million candidates); the ordering keeps its own work buffers
  auto padded = [](size_t bytes) { return (bytes + 255) & ~(size_t)255; };
  const size_t b_rs = padded((size_t)R * sizeof(float)), b_so = padded(off32.size() * sizeof(int32_t)), b_p = padded((size_t)n_pairs * sizeof(float));
  const size_t b_po = do_sort ? padded((size_t)(n_queries + 1) * sizeof(int64_t)) : 0;
  DevBuf block, order, iota, keys, tmp;
  int rc = FFX_OK;
  auto run = [&]() -> int {
    FFX_TRY(block.ensure(b_rs + b_so + (sparse ? 3 : 1) * b_p + b_po));
    char* at = block.as<char>();
    float* d_rs = (float*)at;
    at += b_rs;
    int32_t* d_so = seg_off ? (int32_t*)at : nullptr;
    at += b_so;
    float* d_ff = (float*)at;
    at += b_p;
    float *d_sp = nullptr, *d_in = nullptr;
    if (sparse) {
      d_sp = (float*)at;
      d_in = (float*)(at + b_p);
      at += 2 * b_p;
    }
    int64_t* d_po = (int64_t*)at;
    FFX_TRY(ffx_h2d_now(device, d_rs, row_scores, (size_t)R * sizeof(float)));
    if (seg_off) FFX_TRY(ffx_h2d_now(device, d_so, off32.data(), off32.size() * sizeof(int32_t)));
    if (sparse) FFX_TRY(ffx_h2d_now(device, d_sp, sparse, (size_t)n_pairs * sizeof(float)));
    ffx::reduce_kernel<<<blocks_for(n_pairs, 256), 256>>>(d_rs, d_so, n_pairs, mode, d_sp, (float)alpha, (float)(1.0 - alpha), d_ff, d_in);
    FFX_HIP(hipGetLastError());
    if (do_sort) {
      FFX_TRY(ffx_h2d_now(device, d_po, pair_off, (size_t)(n_queries + 1) * sizeof(int64_t)));
      const float* k = (sparse && !(flags & FFX_FLAG_SORT_FF)) ? d_in : d_ff;
      FFX_TRY(launch_segmented_sort(k, n_pairs, d_po, n_queries, order, iota, keys, tmp, nullptr, longest_segment(pair_off, n_queries)));
    }
    FFX_HIP(hipDeviceSynchronize());
    FFX_TRY(ffx_d2h_now(device, out_ff, d_ff, (size_t)n_pairs * sizeof(float)));
    if (out_interp) FFX_TRY(ffx_d2h_now(device, out_interp, d_in, (size_t)n_pairs * sizeof(float)));
    if (do_sort) FFX_TRY(ffx_d2h_now(device, out_order, order.p, (size_t)n_pairs * sizeof(int32_t)));
    return FFX_OK;
  };
  rc = run();
  for (DevBuf* b : {&block, &order, &iota, &keys, &tmp}) b->release();
  return rc;
}

// One-shot host-to-host form.  The batch is cut into slices of whole queries and the slices
// alternate between two plans on their own streams: while slice c is being scored, the
// candidates of slice c+1 go up and the scores of slice c-1 come down, so with page-locked host
// buffers (ffx_host_alloc / ffx_host_pin) the call costs about one device pass plus one slice of
// transfers.  Results do not depend on the slicing.
// ffx_score for a small batch of single-row candidates (a service scoring one query at a time): everything
// -- uploads out of one page-locked block, validation, expand, gather_dot, interpolation, ordering, the
// downloads back into that block -- is queued on the plan's stream and the host waits ONCE.  The general
// path waits three times (after the queries, after the validation read-back that sizes the row list, after
// the results); here the row list has exactly one row per candidate, so nothing has to come back early,
// and a bad ordinal is reported from the same final read-back.  Returns FFX_OK with *handled = false when
// the batch does not qualify.
static int score_small(ffx_index* idx, const float* host_queries, int64_t n_queries, const int64_t* pair_off, const int32_t* pair_ids,
                       const float* sparse, int mode, double alpha, uint32_t flags, float* out_ff, float* out_interp, int32_t* out_order,
                       int64_t* bad_pair, ffx_stats* stats, bool* handled) {
  *handled = false;
  const int64_t P = pair_off[n_queries];
  const bool one_row = mode == FFX_MODE_PASSAGE || mode == FFX_MODE_FIRSTP || idx->max_doc_rows <= 1;
  if (!one_row || P == 0 || n_queries == 0 || !host_queries || !pair_ids || ffx_debug_variant("no_small_path", nullptr)) return FFX_OK;
  if (mode == FFX_MODE_PASSAGE ? idx->n_psgs == 0 : idx->n_docs == 0) return FFX_OK;  // the general path words that error
  const size_t q_bytes = (size_t)n_queries * idx->qdim * sizeof(float), off_bytes = (size_t)(n_queries + 1) * sizeof(int64_t), p_bytes = (size_t)P * 4;
  const size_t in_bytes = q_bytes + off_bytes + p_bytes * (sparse ? 2 : 1) + 4 * 64;
  const size_t out_bytes = p_bytes * 3 + 64 * 4;
  if (in_bytes + out_bytes > ffx_plan::kStageBytes) return FFX_OK;
  if (!idx->score_plans[0]) FFX_TRY(ffx_plan_create(idx, &idx->score_plans[0]));
  ffx_plan* pl = idx->score_plans[0];
  FFX_HIP(hipSetDevice(idx->device));
  hipStream_t st = pl->stream;
  pl->hstage_used = 0;
  char* s_q = pl->stage(q_bytes);
  char* s_off = pl->stage(off_bytes);
  char* s_ids = pl->stage(p_bytes);
  char* s_sp = sparse ? pl->stage(p_bytes) : nullptr;
  char* b_ff = pl->stage(p_bytes);
  char* b_in = pl->stage(p_bytes);
  char* b_or = pl->stage(p_bytes);
  char* b_flag = pl->stage(sizeof(unsigned long long));
  if (!s_q || !s_off || !s_ids || (sparse && !s_sp) || !b_ff || !b_in || !b_or || !b_flag) return FFX_OK;  // no page-locked block: general path
  const size_t pb = (size_t)P;
  FFX_TRY(pl->queries.ensure(std::max<size_t>(q_bytes, 16)));
  FFX_TRY(pl->pair_off.ensure(off_bytes));
  FFX_TRY(pl->pair_ids.ensure(p_bytes));
  FFX_TRY(pl->ff.ensure(p_bytes));
  FFX_TRY(pl->desc.ensure(pb * sizeof(int2)));
  if (sparse) {
    FFX_TRY(pl->sparse.ensure(p_bytes));
    FFX_TRY(pl->interp.ensure(p_bytes));
  }
  memcpy(s_q, host_queries, q_bytes);
  memcpy(s_off, pair_off, off_bytes);
  memcpy(s_ids, pair_ids, p_bytes);
  if (sparse) memcpy(s_sp, sparse, p_bytes);
  pl->have_pairs = false;
  pl->dense_given = false;
  pl->lazy_pairs = false;
  pl->ran = false;
  pl->n_queries = n_queries;
  pl->n_queries_set = (int)n_queries;
  pl->n_pairs = P;
  pl->n_rows = P;
  pl->mode = mode;
  pl->have_sparse = sparse != nullptr;
  pl->order_base = 0;
  pl->host_pair_off.assign(pair_off, pair_off + n_queries + 1);
  pl->q_hint_valid = false;  // (the hint goes with the offsets it was made from: upload_query_hint makes a new one)
  // page-locked memory is mapped into the device's address space: the kernels read and write the block itself
  if (!pl->small_flag.p) {
    FFX_TRY(pl->small_flag.ensure(sizeof(unsigned long long)));
    FFX_HIP(hipMemsetAsync(pl->small_flag.p, 0xFF, sizeof(unsigned long long), st));
  }
  void* d_block = nullptr;
  FFX_HIP(hipHostGetDevicePointer(&d_block, pl->hstage, 0));
  auto dev = [&](char* host) { return host ? static_cast<char*>(d_block) + (host - pl->hstage) : nullptr; };
  const int64_t q_words = (int64_t)(q_bytes / 4), off_words = (int64_t)(off_bytes / 4);
  const int64_t in_words = q_words + off_words + P * (sparse ? 2 : 1);
  ffx::small_in_kernel<<<(unsigned)std::min<int64_t>(blocks_for(in_words, 256), 2048), 256, 0, st>>>(
      reinterpret_cast<const uint32_t*>(dev(s_q)), reinterpret_cast<const uint32_t*>(dev(s_off)), reinterpret_cast<const int32_t*>(dev(s_ids)),
      reinterpret_cast<const uint32_t*>(dev(s_sp)), q_words, off_words, P, pl->queries.as<uint32_t>(), pl->pair_off.as<uint32_t>(),
      pl->pair_ids.as<int32_t>(), pl->sparse.as<uint32_t>(), mode, idx->doc_off.as<int32_t>(), idx->n_docs, idx->psg_row.as<int32_t>(),
      idx->n_psgs, idx->capacity, pl->small_flag.as<unsigned long long>());
  FFX_HIP(hipGetLastError());
  pl->have_pairs = true;
  pl->derive_valid = false;
  if (const int rc = ffx_plan_run(pl, alpha, flags)) {
    pl->have_pairs = false;
    const std::string keep = g_last_error;
    (void)hipMemsetAsync(pl->small_flag.p, 0xFF, sizeof(unsigned long long), st);  // re-arm: small_out_kernel will not run
    (void)hipStreamSynchronize(st);
    g_last_error = keep;
    return rc;
  }
  const bool want_order = out_order != nullptr;
  ffx::small_out_kernel<<<(unsigned)std::min<int64_t>(blocks_for(P, 256), 1024), 256, 0, st>>>(
      pl->ff.as<uint32_t>(), pl->interp.as<uint32_t>(), pl->order.as<uint32_t>(), P, reinterpret_cast<uint32_t*>(out_ff ? dev(b_ff) : nullptr),
      reinterpret_cast<uint32_t*>(out_interp ? dev(b_in) : nullptr), reinterpret_cast<uint32_t*>(want_order ? dev(b_or) : nullptr),
      pl->small_flag.as<unsigned long long>(), reinterpret_cast<unsigned long long*>(dev(b_flag)));
  FFX_HIP(hipGetLastError());
  FFX_HIP(hipStreamSynchronize(st));  // the one wait
  unsigned long long bad = ~0ull;
  memcpy(&bad, b_flag, sizeof bad);
  if (bad != ~0ull) {
    pl->have_pairs = false;
    if (bad_pair) *bad_pair = (int64_t)bad;
    return fail(FFX_NOT_FOUND, "pair %llu names an ordinal with no rows in the index", bad);
  }
  if (out_ff) memcpy(out_ff, b_ff, p_bytes);
  if (out_interp) memcpy(out_interp, b_in, p_bytes);
  if (want_order) memcpy(out_order, b_or, p_bytes);
  if (stats) FFX_TRY(ffx_plan_stats(pl, stats));
  *handled = true;
  return FFX_OK;
}

// Device memory into ORDINARY host memory without handing that memory to the runtime (which would either page-lock it on
// the fly -- the path this library keeps heap memory off, see the constructor at the top -- or stage it on the null stream, one
// copy at a time for the whole process, at 17 GB/s): 8 MiB pieces go into two page-locked blocks of the plan on the index's
// result stream, and while piece k+1 is on its way the calling thread copies piece k out.
// `lane`: which pair of blocks and stream -- the result arrays of a slice leave side by side, a thread each (fetch_on_thread),
// and up to three slices at a time on three plans: enough for ffx_score's results to leave as fast as they are made.
static int staged_d2h(ffx_plan* pl, int lane, void* dst, const void* src_dev, size_t bytes) {
  if (bytes == 0) return FFX_OK;
  FFX_HIP(hipSetDevice(pl->device));
  if (!pl->out_stage[lane][0]) {
    for (int i = 0; i < 2; ++i) {
      FFX_HIP(hipEventCreateWithFlags(&pl->out_stage_ev[lane][i], hipEventDisableTiming));
      FFX_HIP(hipHostMalloc(reinterpret_cast<void**>(&pl->out_stage[lane][i]), ffx_plan::kOutStageBytes, hipHostMallocDefault));
    }
  }
  // every lane of every plan queues its pieces on the index's ONE result stream (highest priority: a copy that queues behind the
  // next slices' scoring kernels leaves when they are done -- 24.6 ms for the headline batch at normal priority).  A stream per
  // lane (round 4) meant up to fifteen streams on the device's four hardware queues: some shared a queue with the kernel stream.
  hipStream_t st = pl->idx->out_stream;
  size_t piece = ffx_plan::kOutStageBytes;
  if (long long v = 0; ffx_experiment("out_piece_kb", &v) && v >= 64 && (size_t)v * 1024 <= ffx_plan::kOutStageBytes) piece = (size_t)v * 1024;  // A/B
  const size_t n = (bytes + piece - 1) / piece;
  auto issue = [&](size_t k) -> int {
    const size_t at = k * piece, len = std::min(piece, bytes - at);
    FFX_HIP(hipMemcpyAsync(pl->out_stage[lane][k & 1], static_cast<const char*>(src_dev) + at, len, hipMemcpyDeviceToHost, st));
    FFX_HIP(hipEventRecord(pl->out_stage_ev[lane][k & 1], st));
    return FFX_OK;
  };
  FFX_TRY(issue(0));
  for (size_t k = 0; k < n; ++k) {
    if (k + 1 < n) FFX_TRY(issue(k + 1));
    FFX_HIP(hipEventSynchronize(pl->out_stage_ev[lane][k & 1]));
    const size_t at = k * piece, len = std::min(piece, bytes - at);
    memcpy(static_cast<char*>(dst) + at, pl->out_stage[lane][k & 1], len);
  }
  return FFX_OK;
}

// ffx_rank_kept's form of set_queries + set_pairs for one slice: offsets and first-stage scores go up, the dense scores (and the
// carried words) of the slice are copied from where the FFX_FLAG_KEEP_DENSE call left them; the next run starts behind gather_dot
static int plan_set_given(ffx_plan* pl, const int64_t* off, int64_t n_queries, const float* sparse, const float* dev_dense, const int32_t* dev_carry) {
  ffx_index* ix = pl->idx;
  const int64_t P = off[n_queries];
  FFX_HIP(hipSetDevice(ix->device));
  pl->have_pairs = false;
  pl->dense_given = false;
  pl->lazy_pairs = false;
  pl->ran = false;
  pl->n_queries = n_queries;
  pl->n_queries_set = (int)n_queries;  // (no query vector is read)
  pl->n_pairs = pl->n_rows = P;
  pl->have_sparse = sparse != nullptr;
  pl->host_pair_off.assign(off, off + n_queries + 1);
  pl->q_hint_valid = false;  // (the hint goes with the offsets it was made from: upload_query_hint makes a new one)
  const size_t pb = std::max<size_t>((size_t)P, 1) * 4;
  FFX_TRY(pl->pair_off.ensure((size_t)(n_queries + 1) * sizeof(int64_t)));
  FFX_TRY(pl->ff.ensure(pb));
  if (sparse) {
    FFX_TRY(pl->sparse.ensure(pb));
    FFX_TRY(pl->interp.ensure(pb));
  }
  if (dev_carry) FFX_TRY(pl->carry.ensure(pb));
  pl->hstage_used = 0;
  FFX_TRY(plan_h2d(pl, pl->pair_off.p, off, (size_t)(n_queries + 1) * sizeof(int64_t)));
  if (P > 0) {
    if (sparse) FFX_TRY(plan_h2d(pl, pl->sparse.p, sparse, (size_t)P * 4));
    FFX_HIP(hipMemcpyAsync(pl->ff.p, dev_dense, (size_t)P * 4, hipMemcpyDeviceToDevice, pl->stream));
    if (dev_carry) FFX_HIP(hipMemcpyAsync(pl->carry.p, dev_carry, (size_t)P * 4, hipMemcpyDeviceToDevice, pl->stream));
  }
  FFX_HIP(hipStreamSynchronize(pl->stream));  // (as ffx_plan_set_pairs: the run goes to another stream)
  pl->have_pairs = true;
  pl->derive_valid = false;
  pl->dense_given = true;
  return FFX_OK;
}

static uint64_t offsets_hash(const int64_t* off, int64_t n) {  // FNV-1a over the CSR offsets: "the same candidate lists"
  uint64_t h = 1469598103934665603ull;
  for (int64_t i = 0; i <= n; ++i) {
    h ^= (uint64_t)off[i];
    h *= 1099511628211ull;
  }
  return h;
}

static std::atomic<uint64_t> g_kept_tokens{0};

// keep: 0 = ffx_score / ffx_score_ranked; 1 = ... and the dense scores + carried words stay in HBM (FFX_FLAG_KEEP_DENSE);
// 2 = ffx_rank_kept: no candidates, no queries -- every slice starts from the kept dense scores
static int score_impl(ffx_index* idx, const float* host_queries, int64_t n_queries, const int64_t* pair_off,
                      const int32_t* pair_ids, const float* sparse, const int32_t* carry, int mode, double alpha, uint32_t flags, float* out_ff,
                      float* out_interp, int32_t* out_order, float* out_ranked, int32_t* out_carry, int64_t* bad_pair, ffx_stats* stats, int keep = 0) {
  if (bad_pair) *bad_pair = -1;
  if (!idx || !pair_off || n_queries < 0) return fail(FFX_BAD_ARG, "bad argument");
  const bool ranked = out_ranked != nullptr || out_carry != nullptr;
  const bool given = keep == 2;
  if (ranked && !(flags & FFX_FLAG_SORT)) return fail(FFX_BAD_ARG, "the run in final order needs FFX_FLAG_SORT");
  if (out_carry && !carry && !(given && idx->kept_has_carry)) return fail(FFX_BAD_ARG, "out_carry without carry");
  if (pair_off[0] != 0) return fail(FFX_BAD_ARG, "pair_off[0] must be 0");
  for (int64_t q = 0; q < n_queries; ++q)
    if (pair_off[q + 1] < pair_off[q]) return fail(FFX_BAD_ARG, "pair_off not monotone at query %lld", (long long)q);
  const int64_t P = pair_off[n_queries];
  if (P > INT32_MAX - 1) return fail(FFX_BAD_ARG, "more than 2^31-2 pairs in one batch: split the queries");
  if (out_interp && !sparse) return fail(FFX_BAD_ARG, "no sparse scores were set: nothing to interpolate");
  if (out_order && !(flags & FFX_FLAG_SORT)) return fail(FFX_BAD_ARG, "out_order needs FFX_FLAG_SORT");
  if (mode < FFX_MODE_PASSAGE || mode > FFX_MODE_AVEP) return fail(FFX_BAD_ARG, "unknown mode %d", mode);
  if (given && (idx->kept_pairs != P || idx->kept_queries != n_queries || idx->kept_off_hash != offsets_hash(pair_off, n_queries)))
    return fail(FFX_BAD_ARG, "these are not the candidate lists whose dense scores were kept");
  if (keep == 1) {
    FFX_HIP(hipSetDevice(idx->device));
    idx->kept_pairs = idx->kept_queries = -1;  // nothing valid until this call has succeeded
    idx->kept_token = 0;
    FFX_TRY(idx->kept_dense.ensure(std::max<size_t>((size_t)P, 1) * 4));
    if (carry) FFX_TRY(idx->kept_carry.ensure(std::max<size_t>((size_t)P, 1) * 4));
  }
  if (n_queries <= INT32_MAX - 1 && !ranked && keep == 0) {  // small batches of single-row candidates: one submission, one wait
    bool handled = false;
    const int rc_small = score_small(idx, host_queries, n_queries, pair_off, pair_ids, sparse, mode, alpha, flags, out_ff, out_interp, out_order,
                                     bad_pair, stats, &handled);
    if (rc_small != FFX_OK || handled) return rc_small;
  }
  // slices cut at query boundaries: ~4M pairs (a ~2 ms device pass at 768-d fp32) when the caller's
  // buffers are page-locked, ~6M when they are ordinary memory (measured: tools/score_trace.py, HISTORY.md section 8; ordinary buffers: MEASUREMENTS.md section 2)
  int64_t per_slice = 6 << 20;
  {
    hipPointerAttribute_t attr;
    const void* in = given ? (const void*)sparse : (const void*)pair_ids;
    if (in && hipPointerGetAttributes(&attr, in) == hipSuccess && attr.type == hipMemoryTypeHost)
      per_slice = 4 << 20;
    else
      (void)hipGetLastError();  // an ordinary pointer is "invalid value" to the runtime: not an error here
  }
  if (long long v = 0; ffx_debug_variant("score_slice_pairs", &v)) per_slice = std::max<int64_t>(1, (int64_t)v);
  // ... and the first and last slices are smaller (an eighth of a slice, doubling / halving): the device idles
  // while the first slice is uploaded and the caller idles while the last one's results come back
  int64_t ramp_base = ffx_experiment("score_no_ramp", nullptr) ? per_slice : std::max<int64_t>(per_slice / 8, 1);
  if (long long v = 0; ffx_experiment("score_ramp_pairs", &v)) ramp_base = std::max<int64_t>(1, std::min<int64_t>((int64_t)v, per_slice));  // A/B hook: tools/slice_sweep.py
  long long growth_pct = 200;  // how fast the first slices grow (A/B hook)
  (void)ffx_experiment("score_ramp_growth_pct", &growth_pct);
  growth_pct = std::max<long long>(110, std::min<long long>(growth_pct, 400));
  auto slice_target = [&](size_t i, int64_t first_pair) {
    double grown = (double)ramp_base;
    for (size_t j = 0; j < i && grown < (double)per_slice; ++j) grown *= (double)growth_pct / 100.0;
    const int64_t up = (int64_t)std::min<double>(grown, (double)per_slice);
    const int64_t down = std::max<int64_t>(ramp_base, (P - first_pair) / 2);
    return std::min(per_slice, std::min(up, down));
  };
  std::vector<int64_t> cut{0};
  for (int64_t q = 1; q < n_queries; ++q)
    if (pair_off[q + 1] - pair_off[cut.back()] > slice_target(cut.size() - 1, pair_off[cut.back()]) && q > cut.back()) cut.push_back(q);
  cut.push_back(n_queries);
  const int n_slices = (int)cut.size() - 1;
  // Results: into ordinary memory a copy blocks the caller, so slice c-1 is fetched after slice c has been
  // launched and two plans alternate.  Into page-locked memory copies are truly asynchronous: slice c's results
  // are queued on the index's out-stream right behind its kernels and awaited only when its plan is needed
  // again.  With two plans that wait (for slice c-2, i.e. for two slices' kernels sharing the GPU) comes before
  // slice c can even be uploaded, and the device idles for an upload out of every two slices (measured: 25.7 ms
  // for the headline batch); with THREE plans the upload of slice c overlaps the kernels of c-1 and c-2.
  // The kernels of all slices are issued on ONE stream, in order: on a stream per plan they share the GPU, finish
  // together and free their plans together, so the device again idles while the next group is uploaded (23.1 ms).
  // Uploads (and the ordinal check that follows each) go on ONE more stream, plan 0's own, whichever plan the slice
  // uses: the caller awaits each before the next begins, so nothing is lost, and the call then occupies three
  // streams (uploads, kernels, results) instead of five.  The device has four hardware queues for ordinary
  // streams; with a stream per plan two of the five shared one, and when that pair was an upload stream and the
  // kernel stream the upload queued behind a slice's kernels (3.9 ms instead of 0.9 ms for set_pairs: 20.9 ms
  // instead of 18.4 ms for the headline batch, depending on which other streams the process had created).
  bool async_out = false;
  if (n_slices > 1 && !ffx_experiment("score_no_out_stream", nullptr)) {
    const void* any = out_ff ? (const void*)out_ff : out_interp ? (const void*)out_interp : out_order ? (const void*)out_order : out_ranked ? (const void*)out_ranked : (const void*)out_carry;
    hipPointerAttribute_t attr;
    if (any && hipPointerGetAttributes(&attr, any) == hipSuccess && attr.type == hipMemoryTypeHost)
      async_out = true;
    else
      (void)hipGetLastError();
  }
  const bool threaded_out = n_slices > 1 && !async_out && (out_ff || out_interp || out_order || ranked) && !ffx_debug_variant("score_blocking_out", nullptr);
  if ((async_out || threaded_out) && !idx->out_stream) {
    int least = 0, greatest = 0;
    FFX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    FFX_HIP(hipStreamCreateWithPriority(&idx->out_stream, hipStreamNonBlocking, ffx_experiment("score_out_priority_low", nullptr) ? least : greatest));
    for (auto& e : idx->out_done) FFX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  // (ordinary result buffers: FOUR plans (three and five were 0.2 - 0.4 ms slower).  A slice's results leave at the link's pace -- 2 ms for a full slice -- and the short
  // slices a call ends with are scored faster than that: with three plans the third-to-last slice waited for the plan of a full
  // slice whose results were still leaving, and the device idled meanwhile: 1 - 2 ms of the headline batch)
  int n_plans = n_slices > 1 ? (threaded_out ? 4 : async_out ? 3 : 2) : 1;
  if (long long v = 0; n_plans > 2 && ffx_experiment("score_plans", &v)) n_plans = (int)std::max<long long>(3, std::min<long long>(v, ffx_index::kScorePlans));
  for (int i = 0; i < n_plans; ++i)
    if (!idx->score_plans[i]) FFX_TRY(plan_create(idx, &idx->score_plans[i], i == 0));
  if (n_slices > 1 && !idx->score_stream) FFX_HIP(hipStreamCreateWithFlags(&idx->score_stream, hipStreamNonBlocking));
  // where a plan's uploads go; with blocking result copies (two plans alternate, each complete on its stream)
  // the second plan lives on the kernel stream
  hipStream_t home[ffx_index::kScorePlans];
  for (int i = 0; i < ffx_index::kScorePlans; ++i) home[i] = (i == 1 && !async_out && !threaded_out) ? idx->score_stream : idx->score_plans[0]->own_stream;
  struct LentStreams {  // plans 1 and 2 own no stream: they hold none between calls
    ffx_index* idx;
    ~LentStreams() {
      for (int i = 1; i < ffx_index::kScorePlans; ++i)
        if (idx->score_plans[i]) idx->score_plans[i]->stream = nullptr;
    }
  } lent{idx};
  ffx_stats acc;
  memset(&acc, 0, sizeof acc);
  auto add_stats = [&](ffx_plan* pl) -> int {
    if (!stats) return FFX_OK;
    ffx_stats s;
    FFX_TRY(ffx_plan_stats(pl, &s));
    acc.n_queries += s.n_queries;
    acc.n_pairs += s.n_pairs;
    acc.n_rows += s.n_rows;
    acc.algorithmic_bytes += s.algorithmic_bytes;
    acc.ms_expand += s.ms_expand;
    acc.ms_gather_dot += s.ms_gather_dot;
    acc.ms_reduce += s.ms_reduce;
    acc.ms_sort += s.ms_sort;
    acc.ms_total += s.ms_total;
    acc.kernel_variant = s.kernel_variant;
    return FFX_OK;
  };
  auto fetch = [&](int c) -> int {  // blocking: slice c's results into the caller's buffers
    ffx_plan* pl = idx->score_plans[c % n_plans];
    const int64_t p0 = pair_off[cut[c]];
    FFX_TRY(ffx_plan_fetch(pl, out_ff ? out_ff + p0 : nullptr, out_interp ? out_interp + p0 : nullptr,
                           out_order ? out_order + p0 : nullptr));
    if (pl->n_pairs > 0) {
      if (out_ranked) FFX_TRY(ffx_d2h_now(idx->device, out_ranked + p0, pl->ranked.p, (size_t)pl->n_pairs * sizeof(float)));
      if (out_carry) FFX_TRY(ffx_d2h_now(idx->device, out_carry + p0, pl->carry_ranked.p, (size_t)pl->n_pairs * sizeof(int32_t)));
    }
    return add_stats(pl);
  };
  // FFX_SCORE_TRACE=1: host-side timeline of the slices on stderr (us since the call began)
  const bool trace = getenv("FFX_SCORE_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  auto us = [&]() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_begin).count(); };
  const std::function<double()> us_fn = us;
  auto send_out = [&](int c) -> int {  // queue slice c's results behind its kernels
    ffx_plan* pl = idx->score_plans[c % n_plans];
    const int64_t p0 = pair_off[cut[c]];
    const size_t bytes = (size_t)pl->n_pairs * sizeof(float);
    hipStream_t os = idx->out_stream;
    FFX_HIP(hipStreamWaitEvent(os, pl->ev[4], 0));
    if (bytes) {
      if (out_ff) FFX_HIP(hipMemcpyAsync(out_ff + p0, pl->ff.p, bytes, hipMemcpyDeviceToHost, os));
      if (out_interp) FFX_HIP(hipMemcpyAsync(out_interp + p0, pl->interp.p, bytes, hipMemcpyDeviceToHost, os));
      if (out_order) FFX_HIP(hipMemcpyAsync(out_order + p0, pl->order.p, bytes, hipMemcpyDeviceToHost, os));
      if (out_ranked) FFX_HIP(hipMemcpyAsync(out_ranked + p0, pl->ranked.p, bytes, hipMemcpyDeviceToHost, os));
      if (out_carry) FFX_HIP(hipMemcpyAsync(out_carry + p0, pl->carry_ranked.p, bytes, hipMemcpyDeviceToHost, os));
    }
    FFX_HIP(hipEventRecord(idx->out_done[c % n_plans], os));
    return FFX_OK;
  };
  auto retire = [&](int c) -> int {  // slice c's results have landed; its plan is free again
    FFX_HIP(hipEventSynchronize(idx->out_done[c % n_plans]));
    return add_stats(idx->score_plans[c % n_plans]);
  };
  // ordinary result buffers: a copy into them blocks its caller for as long as it takes, so a helper thread per
  // slice does the waiting (three plans, kernels in order on one stream, as above) while this thread uploads
  struct FetchResult {
    int rc = FFX_OK;
    std::string error;
  };
  std::future<FetchResult> fetching[ffx_index::kScorePlans];
  auto fetch_on_thread = [&](int c) -> FetchResult {
    FetchResult r;
    ffx_plan* pl = idx->score_plans[c % n_plans];
    const int64_t p0 = pair_off[cut[c]];
    const size_t bytes = (size_t)pl->n_pairs * sizeof(float);
    auto body = [&]() -> int {
      FFX_HIP(hipSetDevice(idx->device));
      const double w0 = us();
      FFX_HIP(hipEventSynchronize(pl->ev[4]));
      const double w1 = us();
      struct Said {
        bool on; int c; double w0, w1; const std::function<double()>& now;
        ~Said() { if (on) fprintf(stderr, "ffx_score results of slice %d: kernels awaited %.0f .. %.0f, copied out by %.0f\n", c, w0, w1, now()); }
      } said{trace, c, w0, w1, us_fn};
      if (bytes) {
        // every wanted array on a thread of its own (the last one on this one): a thread's memcpy out of the page-locked
        // blocks is what bounds a lane
        struct Job {
          int lane;
          char* dst;
          const char* src;
          size_t len;
        };
        std::vector<Job> jobs;
        // a large array leaves in two halves, a thread each (one thread empties the blocks at ~10 GB/s; three slices' results
        // are on their way at a time, and the slowest decides when its plan is free for the slice three further on)
        const size_t half = bytes >= (size_t(8) << 20) ? ((bytes / 2 + 4095) & ~size_t(4095)) : bytes;
        auto add = [&](int which, void* dst, const void* src) {
          jobs.push_back({2 * which, static_cast<char*>(dst), static_cast<const char*>(src), half});
          if (half < bytes) jobs.push_back({2 * which + 1, static_cast<char*>(dst) + half, static_cast<const char*>(src) + half, bytes - half});
        };
        if (out_ff) add(0, out_ff + p0, pl->ff.p);
        if (out_interp) add(1, out_interp + p0, pl->interp.p);
        if (out_order) add(2, out_order + p0, pl->order.p);
        if (out_ranked) add(3, out_ranked + p0, pl->ranked.p);
        if (out_carry) add(4, out_carry + p0, pl->carry_ranked.p);
        std::vector<std::future<FetchResult>> side;
        auto one = [pl](Job j) {
          FetchResult fr;
          fr.rc = staged_d2h(pl, j.lane, j.dst, j.src, j.len);
          if (fr.rc != FFX_OK) fr.error = g_last_error;
          return fr;
        };
        for (size_t j = 0; j + 1 < jobs.size(); ++j) {
          try {
            side.push_back(std::async(std::launch::async, one, jobs[j]));
          } catch (const std::system_error&) {  // no thread to be had: this one does it
            const FetchResult fr = one(jobs[j]);
            if (fr.rc != FFX_OK) {
              g_last_error = fr.error;
              for (auto& f : side) (void)f.get();
              return fr.rc;
            }
          }
        }
        int rc_all = jobs.empty() ? FFX_OK : staged_d2h(pl, jobs.back().lane, jobs.back().dst, jobs.back().src, jobs.back().len);
        std::string err_all = rc_all != FFX_OK ? g_last_error : std::string();
        for (auto& f : side) {
          const FetchResult fr = f.get();
          if (fr.rc != FFX_OK && rc_all == FFX_OK) {
            rc_all = fr.rc;
            err_all = fr.error;
          }
        }
        if (rc_all != FFX_OK) {
          g_last_error = err_all;
          return rc_all;
        }
      }
      return FFX_OK;
    };
    r.rc = body();
    if (r.rc != FFX_OK) r.error = g_last_error;  // thread-local: carried back to the caller's thread
    return r;
  };
  auto join_fetch = [&](int c) -> int {
    if (!fetching[c % n_plans].valid()) return FFX_OK;
    const FetchResult r = fetching[c % n_plans].get();
    if (r.rc != FFX_OK) {
      g_last_error = r.error;
      return r.rc;
    }
    return add_stats(idx->score_plans[c % n_plans]);
  };
  auto run_all = [&]() -> int {
    std::vector<int64_t> off;
    for (int c = 0; c < n_slices; ++c) {
      ffx_plan* pl = idx->score_plans[c % n_plans];
      const int64_t q0 = cut[c], q1 = cut[c + 1], p0 = pair_off[q0];
      const double t0 = us();
      if (async_out && c >= n_plans) FFX_TRY(retire(c - n_plans));
      if (threaded_out && c >= n_plans) FFX_TRY(join_fetch(c - n_plans));
      const double t0b = us();
      pl->stream = home[c % n_plans];
      if (!given) FFX_TRY(ffx_plan_set_queries(pl, host_queries ? host_queries + (size_t)q0 * idx->qdim : nullptr, q1 - q0));
      const double t1 = us();
      off.resize((size_t)(q1 - q0 + 1));
      for (int64_t q = q0; q <= q1; ++q) off[(size_t)(q - q0)] = pair_off[q] - p0;
      pl->order_base = (int32_t)p0;
      if (ranked) {  // room for the slice in final order; its carried words go up ahead of the lists (set_pairs waits for the stream)
        const size_t sb = std::max<size_t>((size_t)(pair_off[q1] - p0), 1) * 4;
        FFX_TRY(pl->ranked.ensure(sb));
        FFX_TRY(pl->carry_ranked.ensure(sb));
        if (carry) {
          FFX_TRY(pl->carry.ensure(sb));
          if (pair_off[q1] > p0) FFX_TRY(ffx_h2d(idx->device, pl->carry.p, carry + p0, (size_t)(pair_off[q1] - p0) * 4, pl->stream));
        }
      }
      pl->want_ranked = ranked;
      pl->have_carry = carry != nullptr || (given && idx->kept_has_carry);
      int64_t bad = -1;
      const int rc = given ? plan_set_given(pl, off.data(), q1 - q0, sparse ? sparse + p0 : nullptr, idx->kept_dense.as<float>() + p0,
                                            idx->kept_has_carry ? idx->kept_carry.as<int32_t>() + p0 : nullptr)
                           : ffx_plan_set_pairs(pl, off.data(), q1 - q0, pair_ids ? pair_ids + p0 : nullptr, sparse ? sparse + p0 : nullptr, mode, &bad);
      if (rc != FFX_OK) {
        if (bad_pair && bad >= 0) *bad_pair = p0 + bad;
        return rc;
      }
      const double t2 = us();
      if (async_out || threaded_out) pl->stream = idx->score_stream;
      const int rc_run = ffx_plan_run(pl, alpha, flags);
      pl->want_ranked = false;
      if (rc_run == FFX_OK && keep == 1 && pl->n_pairs > 0) {  // behind the slice's kernels, on their stream
        FFX_HIP(hipMemcpyAsync(idx->kept_dense.as<float>() + p0, pl->ff.p, (size_t)pl->n_pairs * 4, hipMemcpyDeviceToDevice, pl->stream));
        if (carry) FFX_HIP(hipMemcpyAsync(idx->kept_carry.as<int32_t>() + p0, pl->carry.p, (size_t)pl->n_pairs * 4, hipMemcpyDeviceToDevice, pl->stream));
        FFX_HIP(hipEventRecord(pl->ev[4], pl->stream));  // "the slice is done" now includes these copies: the plan is not re-used under them
      }
      pl->stream = home[c % n_plans];
      if (rc_run != FFX_OK) return rc_run;
      const double t3 = us();
      if (async_out)
        FFX_TRY(send_out(c));
      else if (threaded_out) {
        try {
          fetching[c % n_plans] = std::async(std::launch::async, fetch_on_thread, c);
        } catch (const std::system_error&) {  // no thread to be had: this one does the waiting
          const FetchResult r = fetch_on_thread(c);
          if (r.rc != FFX_OK) {
            g_last_error = r.error;
            return r.rc;
          }
          FFX_TRY(add_stats(pl));
        }
      } else if (c > 0)
        FFX_TRY(fetch(c - 1));
      if (trace)
        fprintf(stderr, "ffx_score slice %d/%d pairs %lld: begin %.0f wait_plan %.0f set_queries %.0f set_pairs %.0f run %.0f %s %.0f\n", c, n_slices,
                (long long)(pair_off[q1] - p0), t0, t0b - t0, t1 - t0b, t2 - t1, t3 - t2, async_out ? "send_out" : "fetch_prev", us() - t3);
    }
    const double t4 = us();
    int rc = FFX_OK;
    if (async_out) {
      for (int c = std::max(0, n_slices - n_plans); c < n_slices && rc == FFX_OK; ++c) rc = retire(c);
    } else if (threaded_out) {
      for (int c = std::max(0, n_slices - n_plans); c < n_slices && rc == FFX_OK; ++c) rc = join_fetch(c);
    } else {
      rc = fetch(n_slices - 1);
    }
    if (trace) fprintf(stderr, "ffx_score last results %.0f, end %.0f\n", us() - t4, us());
    return rc;
  };
  const int rc = run_all();
  for (auto& f : fetching)
    if (f.valid()) (void)f.get();  // only after a failure: no helper outlives the call
  if (rc != FFX_OK) {  // nothing may still be writing when the caller gets its buffers back
    const std::string keep = g_last_error;
    (void)hipStreamSynchronize(idx->score_plans[0]->own_stream);
    if (idx->score_stream) (void)hipStreamSynchronize(idx->score_stream);
    if (idx->out_stream) (void)hipStreamSynchronize(idx->out_stream);
    g_last_error = keep;
    return rc;
  }
  if (keep == 1) {  // the copies into the kept blocks ran behind the slices' kernels: done before anybody reads them
    for (int i = 0; i < n_plans; ++i) FFX_HIP(hipStreamSynchronize(home[i]));
    if (idx->score_stream) FFX_HIP(hipStreamSynchronize(idx->score_stream));
    idx->kept_pairs = P;
    idx->kept_queries = n_queries;
    idx->kept_has_carry = carry != nullptr;
    idx->kept_off_hash = offsets_hash(pair_off, n_queries);
    idx->kept_token = ++g_kept_tokens;
  }
  if (stats) *stats = acc;
  return FFX_OK;
}

int ffx_score(ffx_index* idx, const float* host_queries, int64_t n_queries, const int64_t* pair_off,
              const int32_t* pair_ids, const float* sparse, int mode, double alpha, uint32_t flags, float* out_ff,
              float* out_interp, int32_t* out_order, int64_t* bad_pair, ffx_stats* stats) {
  return score_impl(idx, host_queries, n_queries, pair_off, pair_ids, sparse, nullptr, mode, alpha, flags, out_ff, out_interp, out_order, nullptr, nullptr,
                    bad_pair, stats);
}

int ffx_score_ranked(ffx_index* idx, const float* host_queries, int64_t n_queries, const int64_t* pair_off, const int32_t* pair_ids,
                     const float* sparse, const int32_t* carry, int mode, double alpha, uint32_t flags, float* out_scores, int32_t* out_carry,
                     int32_t* out_order, int64_t* bad_pair, ffx_stats* stats) {
  if (!out_scores) return fail(FFX_BAD_ARG, "out_scores is NULL");
  return score_impl(idx, host_queries, n_queries, pair_off, pair_ids, sparse, carry, mode, alpha, (flags & ~FFX_FLAG_KEEP_DENSE) | FFX_FLAG_SORT, nullptr, nullptr,
                    out_order, out_scores, out_carry, bad_pair, stats, (flags & FFX_FLAG_KEEP_DENSE) ? 1 : 0);
}

// What the first ffx_score / ffx_score_ranked call of an index would allocate on the way -- the three plans with their
// page-locked staging blocks, the kernel and result streams, every device buffer of a full slice -- made now, so that a
// caller with something else to do first (Index.rerank looks the ranking's ids up: host work) has it done beside that, on
// another thread.  Sizes follow ffx_score's slicing (uniform candidate lists assumed; anything that turns out too small
// simply grows in the call).  Not to be called while another thread uses the index.
int ffx_index_prepare_score(ffx_index* idx, int64_t n_queries, int64_t n_pairs, int32_t has_sparse, int32_t ranked, int32_t page_locked_buffers) {
  if (!idx || n_queries < 0 || n_pairs < 0) return fail(FFX_BAD_ARG, "bad argument");
  if (n_pairs == 0 || n_queries == 0) return FFX_OK;
  FFX_HIP(hipSetDevice(idx->device));
  const int64_t per_slice = page_locked_buffers ? (4 << 20) : (6 << 20);
  const int64_t per_query = (n_pairs + n_queries - 1) / n_queries;
  const bool sliced = n_pairs > per_slice + per_query;
  const int64_t Ps = sliced ? per_slice + 2 * per_query : n_pairs;
  const int64_t Qs = std::min<int64_t>(n_queries, Ps / std::max<int64_t>(per_query, 1) + 2);
  const int n_plans = sliced ? (page_locked_buffers ? 3 : 4) : 1;
  if (sliced && page_locked_buffers && !idx->out_stream) {
    int least = 0, greatest = 0;
    FFX_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    FFX_HIP(hipStreamCreateWithPriority(&idx->out_stream, hipStreamNonBlocking, ffx_experiment("score_out_priority_low", nullptr) ? least : greatest));
    for (auto& e : idx->out_done) FFX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  for (int i = 0; i < n_plans; ++i)
    if (!idx->score_plans[i]) FFX_TRY(plan_create(idx, &idx->score_plans[i], i == 0));
  if (sliced && !idx->score_stream) FFX_HIP(hipStreamCreateWithFlags(&idx->score_stream, hipStreamNonBlocking));
  const size_t pb = (size_t)Ps * 4;
  for (int i = 0; i < n_plans; ++i) {
    ffx_plan* pl = idx->score_plans[i];
    pl->hstage_used = 0;
    (void)pl->stage(16);  // the page-locked staging block of the plan
    pl->hstage_used = 0;
    FFX_TRY(pl->queries.ensure((size_t)Qs * idx->qdim * sizeof(float)));
    FFX_TRY(pl->pair_off.ensure((size_t)(Qs + 1) * sizeof(int64_t)));
    for (DevBuf* b : {&pl->pair_ids, &pl->ff, &pl->order, &pl->iota}) FFX_TRY(b->ensure(pb));
    FFX_TRY(pl->cnt.ensure(pb + 4));
    FFX_TRY(pl->seg_off.ensure(pb + 4));
    FFX_TRY(pl->desc.ensure(2 * pb));
    if (has_sparse) {
      FFX_TRY(pl->sparse.ensure(pb));
      FFX_TRY(pl->interp.ensure(pb));
    }
    if (ranked)
      for (DevBuf* b : {&pl->carry, &pl->ranked, &pl->carry_ranked}) FFX_TRY(b->ensure(pb));
  }
  return FFX_OK;
}

int ffx_index_kept(ffx_index* idx, uint64_t* token, int64_t* n_queries, int64_t* n_pairs) {
  if (!idx) return fail(FFX_BAD_ARG, "NULL index");
  if (token) *token = idx->kept_pairs >= 0 ? idx->kept_token : 0;
  if (n_queries) *n_queries = idx->kept_queries;
  if (n_pairs) *n_pairs = idx->kept_pairs;
  return FFX_OK;
}

int ffx_index_drop_kept(ffx_index* idx) {
  if (!idx) return fail(FFX_BAD_ARG, "NULL index");
  FFX_HIP(hipSetDevice(idx->device));
  idx->kept_pairs = idx->kept_queries = -1;
  idx->kept_token = 0;
  idx->kept_dense.release();
  idx->kept_carry.release();
  return FFX_OK;
}

int ffx_rank_kept(ffx_index* idx, uint64_t token, int64_t n_queries, const int64_t* pair_off, const float* sparse, double alpha, uint32_t flags,
                  float* out_scores, int32_t* out_carry, int32_t* out_order, ffx_stats* stats) {
  if (!idx || !out_scores) return fail(FFX_BAD_ARG, "bad argument");
  if (token == 0 || idx->kept_pairs < 0 || token != idx->kept_token) return fail(FFX_BAD_ARG, "no dense scores are kept under this token");
  return score_impl(idx, nullptr, n_queries, pair_off, nullptr, sparse, nullptr, FFX_MODE_PASSAGE, alpha, (flags & ~FFX_FLAG_KEEP_DENSE) | FFX_FLAG_SORT, nullptr,
                    nullptr, out_order, out_scores, out_carry, nullptr, stats, 2);
}

// ---- page-locked host memory ------------------------------------------------------------------
int ffx_host_alloc(size_t bytes, void** out) {
  if (!out) return fail(FFX_BAD_ARG, "out is NULL");
  struct Say {
    size_t bytes;
    void** out;
    ~Say() {
      if (ffx_trace_host()) fprintf(stderr, "\nffx trace: host_alloc %zu bytes -> %p\n", bytes, *out);
    }
  } say{bytes, out};
  *out = nullptr;
  hipError_t e = hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault);
  if (e != hipSuccess) {
    *out = nullptr;
    return fail(FFX_OOM, "hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
  }
  return FFX_OK;
}

int ffx_host_free(void* p) {
  if (p && ffx_trace_host()) fprintf(stderr, "\nffx trace: host_free %p\n", p);
  if (p) FFX_HIP(hipHostFree(p));
  return FFX_OK;
}

// A refusal (RLIMIT_MEMLOCK in a container, a mapping that cannot be read) is an answer, not a fault: callers fall back
// to ordinary copies on the same thread, so the runtime's thread-local last error is cleared before returning -- the next
// launch check (FFX_HIP(hipGetLastError())) must not report this stale failure as its own.
int ffx_host_pin(void* p, size_t bytes) {
  if (!p || bytes == 0) return fail(FFX_BAD_ARG, "bad argument");
  if (ffx_trace_host()) fprintf(stderr, "\nffx trace: host_pin %p..%p\n", p, (char*)p + bytes);
  const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(e == hipErrorOutOfMemory ? FFX_OOM : FFX_HIP_ERROR, "hipHostRegister(%zu bytes): %s", bytes, hipGetErrorString(e));
  }
  return FFX_OK;
}

int ffx_host_unpin(void* p) {
  if (!p) return fail(FFX_BAD_ARG, "bad argument");
  if (ffx_trace_host()) fprintf(stderr, "\nffx trace: host_unpin %p\n", p);
  const hipError_t e = hipHostUnregister(p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(FFX_HIP_ERROR, "hipHostUnregister: %s", hipGetErrorString(e));
  }
  return FFX_OK;
}

}  // extern "C"
