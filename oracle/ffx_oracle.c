/*
 * ffx_oracle.c -- CPU restatement of the Fast-Forward re-ranking arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it; the product package (fast-forward-indexes_amd/) never does.
 *
 * Parity status: PINNED.  Every function here is checked against
 *   (a) the reference's own known-answer tests (tests/test_index.py:135-200,
 *       :273-333, tests/test_ranking.py:157-188) and
 *   (b) outputs of the reference itself, imported in the build container and
 *       run on seeded inputs by oracle/gen_golden.py (fixtures in tests/golden/).
 *
 * Each function cites the reference lines (relative to /root/reference) whose
 * arithmetic it restates.  Nothing here is copied from the reference: the
 * reference is pandas/NumPy calls; this is the arithmetic those calls perform,
 * written out for integer CSR inputs.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math: an FMA
 * or a re-associated sum would break bit-exactness).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define FFO_MODE_PASSAGE 1
#define FFO_MODE_MAXP 2
#define FFO_MODE_FIRSTP 3
#define FFO_MODE_AVEP 4

#define FFO_OK 0
#define FFO_BAD_ARG (-1)
#define FFO_NOT_FOUND (-5)

/* ---- synthetic data: counter-based generator (identical to the HIP and
 * NumPy versions; all operations exact except the final, correctly rounded,
 * multiply by `scale`) ------------------------------------------------------ */
static inline uint64_t ffo_mix(uint64_t idx, uint64_t seed) {
  uint64_t z = idx + seed * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static inline float ffo_synth_value(uint64_t idx, uint64_t seed, float scale) {
  uint32_t m = (uint32_t)(ffo_mix(idx, seed) >> 40); /* 24 bits */
  float u = (float)m * 0x1p-23f - 1.0f;              /* exact, in [-1, 1) */
  return u * scale;
}

void ffo_synth_rows(uint64_t seed, float scale, int64_t first_row, int64_t n_rows, int64_t dim,
                    float* out) {
  for (int64_t r = 0; r < n_rows; ++r)
    for (int64_t c = 0; c < dim; ++c)
      out[r * dim + c] = ffo_synth_value((uint64_t)((first_row + r) * dim + c), seed, scale);
}

/* the same generator for an arbitrary list of global rows (bench: rebuild on the host only the
 * rows a CPU-baseline sample touches) */
void ffo_synth_row_list(uint64_t seed, float scale, const int64_t* rows, int64_t n_rows, int64_t dim,
                        float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n_rows; ++r)
    for (int64_t c = 0; c < dim; ++c)
      out[r * dim + c] = ffo_synth_value((uint64_t)(rows[r] * dim + c), seed, scale);
}

/* ---- row score ------------------------------------------------------------
 * src/fast_forward/index/base.py:303   ff_score = np.sum(q_reps * d_reps, axis=1)
 * fp32 x fp32: every product is rounded to fp32 on its own (no FMA), then the
 * row is reduced by NumPy's pairwise summation (numpy/_core/src/umath/
 * loops_utils.h.src, pairwise_sum_@TYPE@, numpy 2.2.6; PW_BLOCKSIZE = 128):
 *   n < 8      : sequential, starting from 0
 *   n <= 128   : 8 strided accumulators, tree-combined, then the n%8 tail
 *   otherwise  : split at n2 = n/2 - (n/2 % 8), recurse on both halves.
 * The reduction then adds the result to the add-identity 0 (a no-op). */
static float ffo_pairwise_prod(const float* q, const float* v, int64_t n) {
  if (n < 8) {
    float res = -0.0f; /* numpy starts from -0 so that a sum of -0s stays -0 */
    for (int64_t i = 0; i < n; ++i) {
      float p = q[i] * v[i];
      res += p;
    }
    return res;
  }
  if (n <= 128) {
    float r[8];
    for (int j = 0; j < 8; ++j) r[j] = q[j] * v[j];
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) {
        float p = q[i + j] * v[i + j];
        r[j] += p;
      }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) {
      float p = q[i] * v[i];
      res += p;
    }
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  float a = ffo_pairwise_prod(q, v, n2);
  float b = ffo_pairwise_prod(q + n2, v + n2, n - n2);
  return a + b;
}

float ffo_dot_f32(const float* q, const float* v, int64_t d) { return ffo_pairwise_prod(q, v, d); }

/* Scores of R gathered rows: row r pairs index row row_ids[r] with query
 * row_q[r].  Restates the two gathers + product-sum of base.py:301-303
 * without materialising q_reps / d_reps. */
void ffo_row_scores(const float* queries, const float* vectors, int64_t d, const int32_t* row_q,
                    const int32_t* row_ids, int64_t n_rows, float* out) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n_rows; ++r)
    out[r] = ffo_pairwise_prod(queries + (int64_t)row_q[r] * d, vectors + (int64_t)row_ids[r] * d, d);
}

/* ---- aggregation over one (id, q_no) group ---------------------------------
 * src/fast_forward/index/base.py:306-312  groupby(["id","q_no"]).aggregate(op)
 *   MAXP -> "max"    plain fp32 max over the group's rows
 *   AVEP -> "mean"   pandas group_mean: Kahan-compensated fp32 sum in row
 *                    order, then ONE fp32 divide by the fp32 count
 *                    (pandas/_libs/groupby.pyx group_mean, pandas 2.3.3)
 *   else -> "first"  first row of the group in frame order (= insertion order
 *                    of the doc's passages, util.py:29-41)             */
static float ffo_aggregate_one(const float* s, int64_t n, int mode) {
  if (mode == FFO_MODE_MAXP) {
    float m = s[0];
    for (int64_t i = 1; i < n; ++i)
      if (s[i] > m) m = s[i];
    return m;
  }
  if (mode == FFO_MODE_AVEP) {
    float sum = 0.f, comp = 0.f, cnt = 0.f;
    for (int64_t i = 0; i < n; ++i) {
      cnt += 1.f;
      float y = s[i] - comp;
      float t = sum + y;
      comp = (t - sum) - y;
      if (comp != comp) comp = 0.f; /* inf - inf guard, as pandas does */
      sum = t;
    }
    return sum / cnt;
  }
  return s[0];
}

void ffo_aggregate(const float* row_scores, const int64_t* seg_off, int64_t n_pairs, int mode,
                   float* out) {
  for (int64_t p = 0; p < n_pairs; ++p)
    out[p] = ffo_aggregate_one(row_scores + seg_off[p], seg_off[p + 1] - seg_off[p], mode);
}

/* ---- interpolation -----------------------------------------------------------
 * src/fast_forward/ranking.py:319  score = alpha*score + (1-alpha)*score_other
 * on float32 columns with a Python-float alpha: both weights are cast to fp32
 * ((1-alpha) is formed in double first), each product is rounded to fp32, then
 * one fp32 add.  Same expression at index/base.py:377-379 (early stopping). */
void ffo_interpolate(const float* sparse, const float* dense, int64_t n, double alpha, float* out) {
  const float a = (float)alpha;
  const float b = (float)(1.0 - alpha);
  for (int64_t i = 0; i < n; ++i) {
    float x = a * sparse[i];
    float y = b * dense[i];
    out[i] = x + y;
  }
}

/* ---- the whole path on CSR inputs ------------------------------------------
 * Restates Index._compute_scores (index/base.py:279-314) + get_indices
 * (index/util.py:12-42) for integer ids:
 *   pair p of query q (pair_off[q] <= p < pair_off[q+1]) names ordinal
 *   pair_ids[p]: a document ordinal (MAXP/FIRSTP/AVEP; rows doc_rows[doc_off[o]
 *   .. doc_off[o+1]) in insertion order) or a passage ordinal (PASSAGE; row
 *   psg_row[o]).  An ordinal with no rows is the reference's IndexError
 *   (util.py:38-39): returns FFO_NOT_FOUND and the pair position in *bad_pair.
 * out_ff[p]     = aggregated dense score     (base.py:303-312)
 * out_interp[p] = interpolation with sparse  (ranking.py:319), if both given */
int ffo_score(const float* vectors, int64_t n_rows, int64_t d, const int64_t* doc_off,
              const int32_t* doc_rows, int64_t n_docs, const int32_t* psg_row, int64_t n_psgs,
              const float* queries, int64_t n_queries, const int64_t* pair_off,
              const int32_t* pair_ids, int mode, const float* sparse, double alpha, float* out_ff,
              float* out_interp, int64_t* bad_pair) {
  if (mode < FFO_MODE_PASSAGE || mode > FFO_MODE_AVEP) return FFO_BAD_ARG;
  const int64_t n_pairs = pair_off[n_queries];
  for (int64_t p = 0; p < n_pairs; ++p) {
    int32_t o = pair_ids[p];
    int ok;
    if (mode == FFO_MODE_PASSAGE)
      ok = o >= 0 && o < n_psgs && psg_row[o] >= 0 && psg_row[o] < n_rows;
    else
      ok = o >= 0 && o < n_docs && doc_off[o + 1] > doc_off[o];
    if (!ok) {
      if (bad_pair) *bad_pair = p;
      return FFO_NOT_FOUND;
    }
  }
  int err = 0;
#pragma omp parallel
  {
    float stack_buf[64];
    for (int64_t q = 0; q < n_queries; ++q) {
      const float* qv = queries + q * d;
#pragma omp for schedule(static) nowait
      for (int64_t p = pair_off[q]; p < pair_off[q + 1]; ++p) {
        int32_t o = pair_ids[p];
        int64_t n;
        const int32_t* rows;
        if (mode == FFO_MODE_PASSAGE) {
          rows = psg_row + o;
          n = 1;
        } else {
          rows = doc_rows + doc_off[o];
          n = doc_off[o + 1] - doc_off[o];
          if (mode == FFO_MODE_FIRSTP) n = 1;
        }
        float* s = n <= 64 ? stack_buf : (float*)malloc((size_t)n * sizeof(float));
        if (!s) {
#pragma omp atomic write
          err = 1;
          continue;
        }
        for (int64_t i = 0; i < n; ++i) s[i] = ffo_pairwise_prod(qv, vectors + (int64_t)rows[i] * d, d);
        out_ff[p] = ffo_aggregate_one(s, n, mode);
        if (s != stack_buf) free(s);
      }
    }
  }
  if (err) return FFO_BAD_ARG;
  if (sparse && out_interp) ffo_interpolate(sparse, out_ff, n_pairs, alpha, out_interp);
  return FFO_OK;
}

/* Per-query ordering: src/fast_forward/ranking.py:115-117 sorts by score
 * descending inside each query.  pandas' quicksort leaves the order of exact
 * ties unspecified; the build fixes it as "ascending position in the input
 * candidate list" (stable).  out_order[p] = position (within the whole pair
 * array) of the candidate ranked at p. */
typedef struct {
  float s;
  int64_t pos;
} ffo_kv;

static int ffo_cmp_desc(const void* a, const void* b) {
  const ffo_kv* x = (const ffo_kv*)a;
  const ffo_kv* y = (const ffo_kv*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  return (x->pos > y->pos) - (x->pos < y->pos);
}

int ffo_order(const float* scores, int64_t n_queries, const int64_t* pair_off, int64_t* out_order) {
  for (int64_t q = 0; q < n_queries; ++q) {
    int64_t lo = pair_off[q], n = pair_off[q + 1] - lo;
    ffo_kv* kv = (ffo_kv*)malloc((size_t)(n ? n : 1) * sizeof(ffo_kv));
    if (!kv) return FFO_BAD_ARG;
    for (int64_t i = 0; i < n; ++i) {
      kv[i].s = scores[lo + i];
      kv[i].pos = lo + i;
    }
    qsort(kv, (size_t)n, sizeof(ffo_kv), ffo_cmp_desc);
    for (int64_t i = 0; i < n; ++i) out_order[lo + i] = kv[i].pos;
    free(kv);
  }
  return FFO_OK;
}
