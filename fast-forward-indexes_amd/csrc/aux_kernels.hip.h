// aux_kernels.hip.h -- the small kernels either side of gather_dot (all P- or R-sized streams,
// <1 % of the bytes of the gather): candidate expansion, segment aggregation + interpolation,
// row copy-out, synthetic fill.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <rocprim/block/block_radix_sort.hpp>

#include "../../include/ffx.h"

namespace ffx {

// ---- get_indices (src/fast_forward/index/util.py:12-42) on ordinals ----------------------
// cnt[p] = number of index rows candidate p contributes; an ordinal that names no row is the
// reference's IndexError: the smallest such pair position lands in *bad_pair.
__global__ void __launch_bounds__(256)
count_rows_kernel(const int32_t* __restrict__ pair_ids, int64_t n_pairs, int mode,
                  const int32_t* __restrict__ doc_off, int64_t n_docs,
                  const int32_t* __restrict__ psg_row, int64_t n_psgs, int64_t n_index_rows,
                  int32_t* __restrict__ cnt, unsigned long long* __restrict__ bad_pair) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p > n_pairs) return;
  if (p == n_pairs) {  // sentinel so that an exclusive scan of n_pairs+1 counts ends with R
    cnt[p] = 0;
    return;
  }
  const int32_t o = pair_ids[p];
  int32_t c = 0;
  if (mode == FFX_MODE_PASSAGE) {
    if (o >= 0 && o < n_psgs) {
      const int32_t r = psg_row[o];
      c = (r >= 0 && r < n_index_rows) ? 1 : 0;
    }
  } else if (o >= 0 && o < n_docs) {
    c = doc_off[o + 1] - doc_off[o];
    if (mode == FFX_MODE_FIRSTP && c > 1) c = 1;
  }
  cnt[p] = c;
  if (c == 0) atomicMin(bad_pair, (unsigned long long)p);
}

// ---- ffx_score's single-submission path (small batches, one row per candidate) ---------------
// The batch arrives in a page-locked host block the device can address.  One kernel moves it into the
// plan's device buffers (four copies on the copy engine cost ~10 us each at these sizes; this is one
// launch) and validates the ordinals on the way, as count_rows_kernel does: the smallest bad position lands in
// *bad_pair, which holds ~0 between calls.  A bad ordinal is replaced by 0 so that the ordinal -> row tables
// are read in bounds (the row itself is clamped where it is looked up); the call fails once *bad_pair has
// been read back, which happens together with the results.
__global__ void __launch_bounds__(256)
small_in_kernel(const uint32_t* __restrict__ h_queries, const uint32_t* __restrict__ h_pair_off, const int32_t* __restrict__ h_pair_ids,
                const uint32_t* __restrict__ h_sparse, int64_t q_words, int64_t off_words, int64_t n_pairs,
                uint32_t* __restrict__ queries, uint32_t* __restrict__ pair_off, int32_t* __restrict__ pair_ids, uint32_t* __restrict__ sparse,
                int mode, const int32_t* __restrict__ doc_off, int64_t n_docs, const int32_t* __restrict__ psg_row, int64_t n_psgs,
                int64_t n_index_rows, unsigned long long* __restrict__ bad_pair) {
  const int64_t total = q_words + off_words + n_pairs * (h_sparse ? 2 : 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t j = i;
    if (j < q_words) {
      queries[j] = h_queries[j];
      continue;
    }
    j -= q_words;
    if (j < off_words) {
      pair_off[j] = h_pair_off[j];
      continue;
    }
    j -= off_words;
    if (j >= n_pairs) {
      sparse[j - n_pairs] = h_sparse[j - n_pairs];
      continue;
    }
    int32_t o = h_pair_ids[j];
    bool ok = false;
    if (mode == FFX_MODE_PASSAGE) {
      if (o >= 0 && o < n_psgs) {
        const int32_t r = psg_row[o];
        ok = r >= 0 && r < n_index_rows;
      }
    } else if (o >= 0 && o < n_docs) {
      ok = doc_off[o + 1] > doc_off[o];
    }
    if (!ok) {
      atomicMin(bad_pair, (unsigned long long)j);
      o = 0;
    }
    pair_ids[j] = o;
  }
}

// ... and one kernel writes the results into the host block (each source may be absent) and re-arms *bad_pair.
__global__ void __launch_bounds__(256)
small_out_kernel(const uint32_t* __restrict__ ff, const uint32_t* __restrict__ interp, const uint32_t* __restrict__ order, int64_t n_pairs,
                 uint32_t* __restrict__ h_ff, uint32_t* __restrict__ h_interp, uint32_t* __restrict__ h_order,
                 unsigned long long* __restrict__ bad_pair, unsigned long long* __restrict__ h_bad_pair) {
  const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = t0; i < n_pairs; i += (int64_t)gridDim.x * blockDim.x) {
    if (h_ff) h_ff[i] = ff[i];
    if (h_interp) h_interp[i] = interp[i];
    if (h_order) h_order[i] = order[i];
  }
  if (t0 == 0) {
    *h_bad_pair = *bad_pair;
    *bad_pair = ~0ull;
  }
}

// out[i] = src[index[i]]
// The run in FINAL order (ranking.py:115-117 applied, not just described): position i of the per-query order holds candidate
// order[i] - base of the slice; its key (the score the order was made by) and its carried word (whatever the caller numbered
// the candidates with: id codes of a Ranking) land at i.  Reads of a query's block come out of the cache; writes are coalesced.
__global__ void __launch_bounds__(256)
rank_take_kernel(const float* __restrict__ keys, const int32_t* __restrict__ carry, const int32_t* __restrict__ order, int32_t base, int64_t n,
                 float* __restrict__ out_keys, int32_t* __restrict__ out_carry) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t j = order[i] - base;
  out_keys[i] = keys[j];
  if (carry) out_carry[i] = carry[j];
}

__global__ void __launch_bounds__(256)
take_f32_kernel(const float* __restrict__ src, const int32_t* __restrict__ index, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = src[index[i]];
}

// *bad = smallest i with v[i] outside 0 .. limit-1 (stays ~0 when all are inside)
__global__ void __launch_bounds__(256)
check_range_kernel(const int32_t* __restrict__ v, int64_t n, int32_t limit, unsigned long long* __restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && (uint32_t)v[i] >= (uint32_t)limit) atomicMin(bad, (unsigned long long)i);
}

// query number of pair p: the q with pair_off[q] <= p < pair_off[q+1].  First-stage runs give
// (nearly) every query the same number of candidates, so the proportional guess is almost
// always right and costs two independent loads; a miss falls back to binary search.
__device__ __forceinline__ int query_of_pair(const int64_t* __restrict__ pair_off, int n_queries,
                                             int64_t p, double queries_per_pair) {
  int q = (int)((double)p * queries_per_pair);
  q = q < 0 ? 0 : (q > n_queries - 1 ? n_queries - 1 : q);
  if (pair_off[q] <= p && p < pair_off[q + 1]) return q;
  int lo = 0, hi = n_queries;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pair_off[mid] <= p) lo = mid; else hi = mid;
  }
  return lo;
}

// desc[seg_off[p] + i] = {i-th row of candidate p, query number of p}
__global__ void __launch_bounds__(256)
expand_kernel(const int32_t* __restrict__ pair_ids, int64_t n_pairs, int mode,
              const int64_t* __restrict__ pair_off, int n_queries, double queries_per_pair,
              const int32_t* __restrict__ doc_off, const int32_t* __restrict__ doc_rows,
              const int32_t* __restrict__ psg_row, const int32_t* __restrict__ seg_off,
              int2* __restrict__ desc) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const int q = query_of_pair(pair_off, n_queries, p, queries_per_pair);
  const int32_t o = pair_ids[p];
  const int32_t s = seg_off[p];
  const int32_t c = seg_off[p + 1] - s;
  if (mode == FFX_MODE_PASSAGE) {
    desc[s] = make_int2(psg_row[o], q);
  } else {
    const int32_t* rows = doc_rows + doc_off[o];
    for (int32_t i = 0; i < c; ++i) desc[s + i] = make_int2(rows[i], q);
  }
}

// For the gather_dot kernels that derive their rows themselves (gather_dot.hip.h, Derive; documents of several passages,
// MAXP / AVEP): per candidate where its document's rows begin in doc_rows, per TILE of the row list the candidate its first row
// belongs to and that candidate's query.  4 bytes per candidate and 8 per tile; made once per candidate list.
__global__ void __launch_bounds__(256)
tile_index_kernel(const int32_t* __restrict__ pair_ids, int64_t n_pairs, const int64_t* __restrict__ pair_off, int n_queries,
                  double queries_per_pair, const int32_t* __restrict__ doc_off, const int32_t* __restrict__ seg_off, int tile,
                  int32_t* __restrict__ pair_first, int2* __restrict__ tile_tab) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const int32_t s = seg_off[p], e = seg_off[p + 1];
  pair_first[p] = doc_off[pair_ids[p]];
  int64_t t = ((int64_t)s + tile - 1) / tile;
  if (t * tile < e) {
    const int q = query_of_pair(pair_off, n_queries, p, queries_per_pair);
    for (; t * tile < e; ++t) tile_tab[t] = make_int2((int)p, q);
  }
}

#ifdef FFX_EXPERIMENTS
// FFX_DEBUG_VARIANT=doc_major (ffx_plan_run): the row list in row order and back
__global__ void __launch_bounds__(256) dm_prepare_kernel(const int2* __restrict__ desc, int64_t n, uint32_t* __restrict__ keys, int32_t* __restrict__ idx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    keys[i] = (uint32_t)desc[i].x;
    idx[i] = (int32_t)i;
  }
}
__global__ void __launch_bounds__(256) dm_gather_kernel(const int2* __restrict__ desc, const int32_t* __restrict__ perm, int64_t n, int2* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = desc[perm[i]];
}
__global__ void __launch_bounds__(256) dm_scatter_kernel(const float* __restrict__ src, const int32_t* __restrict__ perm, int64_t n, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[perm[i]] = src[i];
}
#endif

// single-row candidates (FIRSTP, PASSAGE, or every document has one passage): no scan needed.
// row_of[o] is the one row of ordinal o (first row of the document / row of the passage);
// row_of == nullptr means ordinal == row (ids were added one per row, in order).
__global__ void __launch_bounds__(256)
expand_single_kernel(const int32_t* __restrict__ pair_ids, int64_t n_pairs,
                     const int64_t* __restrict__ pair_off, int n_queries, double queries_per_pair,
                     const int32_t* __restrict__ row_of, int64_t n_index_rows, int2* __restrict__ desc) {
  // four candidates per thread: one 16-byte load, two 16-byte stores (the kernel moves 12 bytes per candidate and
  // nothing else; one candidate per thread left it at half the rate the memory system gives wide accesses)
  const int64_t p0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (p0 >= n_pairs) return;
  // validated ordinals always give a row of the store; the clamp is for ffx_score's single-submission path,
  // where the verdict on a bad ordinal is read back only after the kernels behind this one have run
  auto row_at = [&](int32_t o) {
    const int32_t row = row_of ? row_of[o] : o;
    return (uint32_t)row < (uint64_t)n_index_rows ? row : 0;
  };
  if (p0 + 4 <= n_pairs) {
    const int4 o = *reinterpret_cast<const int4*>(pair_ids + p0);  // p0 is a multiple of 4 and the buffer comes from hipMalloc: aligned
    const int q0 = query_of_pair(pair_off, n_queries, p0, queries_per_pair);
    const int64_t q_end = pair_off[q0 + 1];  // the next three usually belong to the same query
    const int q1 = p0 + 1 < q_end ? q0 : query_of_pair(pair_off, n_queries, p0 + 1, queries_per_pair);
    const int q2 = p0 + 2 < q_end ? q0 : query_of_pair(pair_off, n_queries, p0 + 2, queries_per_pair);
    const int q3 = p0 + 3 < q_end ? q0 : query_of_pair(pair_off, n_queries, p0 + 3, queries_per_pair);
    int4* out = reinterpret_cast<int4*>(desc + p0);
    out[0] = make_int4(row_at(o.x), q0, row_at(o.y), q1);
    out[1] = make_int4(row_at(o.z), q2, row_at(o.w), q3);
    return;
  }
  for (int64_t p = p0; p < n_pairs; ++p) desc[p] = make_int2(row_at(pair_ids[p]), query_of_pair(pair_off, n_queries, p, queries_per_pair));
}

// ---- groupby(["id","q_no"]).aggregate(op) (index/base.py:306-312) + ranking.py:319 ----------
// One thread per candidate; its rows are contiguous in rs, so no atomics.
//   MAXP  plain fp32 max            AVEP  pandas group_mean: Kahan fp32 sum, one fp32 divide
//   FIRSTP / PASSAGE  first row
// interp = f32(f32(alpha)*sparse) + f32(f32(1-alpha)*dense): two rounded multiplies, one add
// (no FMA: __f*_rn intrinsics are never contracted).
__device__ __forceinline__ float interp_one(float a, float b, float sparse, float dense) {
  return __fadd_rn(__fmul_rn(a, sparse), __fmul_rn(b, dense));
}

__global__ void __launch_bounds__(256)
reduce_kernel(const float* __restrict__ rs, const int32_t* __restrict__ seg_off, int64_t n_pairs,
              int mode, const float* __restrict__ sparse, float a, float b,
              float* __restrict__ ff, float* __restrict__ interp) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const int32_t s = seg_off ? seg_off[p] : (int32_t)p, e = seg_off ? seg_off[p + 1] : s + 1;  // no offsets: one row each
  float out;
  if (mode == FFX_MODE_MAXP) {
    out = rs[s];
    for (int32_t i = s + 1; i < e; ++i) {
      const float x = rs[i];
      out = x > out ? x : out;
    }
  } else if (mode == FFX_MODE_AVEP) {
    float sum = 0.f, comp = 0.f, cnt = 0.f;
    for (int32_t i = s; i < e; ++i) {
      cnt = __fadd_rn(cnt, 1.f);
      const float y = __fsub_rn(rs[i], comp);
      const float t = __fadd_rn(sum, y);
      comp = __fsub_rn(__fsub_rn(t, sum), y);
      if (comp != comp) comp = 0.f;
      sum = t;
    }
    out = __fdiv_rn(sum, cnt);
  } else {
    out = rs[s];
  }
  ff[p] = out;
  if (sparse != nullptr) interp[p] = interp_one(a, b, sparse[p], out);
}

// Row sum of a long row from the sums of its segments (gather_dot_exact_anyd_kernel, one launch per segment):
// part[g * n_rows + r] is segment g of row r; the steps "sum[a] += sum[b]" replay NumPy's recursion above the
// segments (the total ends in segment 0).  One thread per row, its partial sums in its own column of LDS.
__global__ void __launch_bounds__(256)
combine_tree_kernel(const float* __restrict__ part, int n_rows, int n_segments, const int* __restrict__ steps, int n_steps,
                    float* __restrict__ out) {
  extern __shared__ float seg_sum[];  // [n_segments][256]
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rows) return;
  for (int g = 0; g < n_segments; ++g) seg_sum[g * 256 + threadIdx.x] = part[(size_t)g * n_rows + r];
  for (int t = 0; t < n_steps; ++t) {
    const int a = steps[2 * t], b = steps[2 * t + 1];
    seg_sum[a * 256 + threadIdx.x] = __fadd_rn(seg_sum[a * 256 + threadIdx.x], seg_sum[b * 256 + threadIdx.x]);
  }
  out[r] = seg_sum[threadIdx.x];
}

// ---- early stopping (index/base.py:316-387) on the device ---------------------------------------------
// State: active[q] (the query is still being scored), full-length ff / interp (NaN = not scored).
// Candidates of a query stand in first-stage rank order, so "ranks a..b-1" is a slice of its list.

// order-preserving image of a float: larger float <=> larger unsigned
__device__ __forceinline__ unsigned float_image(float x) {
  const unsigned u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float image_float(unsigned k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// The continue-criterion of base.py:350-358 for every query at once, one block per query: the
// cutoff-th best interpolated score so far (the worst one if fewer were scored) must be below
//   alpha * (sparse score of the last scored candidate) + (1 - alpha) * (best dense score so far)
// The reference's expression multiplies a Python float by a np.float32 SCALAR, so what it computes depends on
// NumPy's scalar promotion (pinned by tests/golden/early_stopping_np1.npz / _np2.npz, both produced by the reference):
//   bound_f64 (default)  NumPy < 2, the reference's pinned range: both products and the sum in float64 -- the
//                        cutoff-th score (float32) is compared with a double
//   !bound_f64           NumPy >= 2 (NEP 50): python_float * np.float32 stays float32, everything in float32
// The k-th largest comes from a 4-pass radix select on the float images: exact, independent of cutoff.
__global__ void __launch_bounds__(256)
es_criterion_kernel(const float* __restrict__ interp, const float* __restrict__ ff, const float* __restrict__ sparse,
                    const int64_t* __restrict__ pair_off, int scored_depth, int cutoff, float a, float b, double a64, double b64,
                    int bound_f64, int32_t* __restrict__ active) {
  __shared__ unsigned hist[256];
  __shared__ unsigned s_prefix, s_k;
  __shared__ float s_max[4];
  const int q = blockIdx.x;
  if (!active[q]) return;  // block-uniform
  const int64_t lo = pair_off[q];
  const int64_t len = pair_off[q + 1] - lo;
  const int n = (int)(len < scored_depth ? len : scored_depth);
  if (n == 0) {  // nothing scored: the reference's filter never sees the query, so it is not continued
    if (threadIdx.x == 0) active[q] = 0;
    return;
  }
  if (threadIdx.x == 0) {
    s_prefix = 0;
    s_k = (unsigned)(cutoff < n ? cutoff : n);
  }
  unsigned mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    hist[threadIdx.x] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix;
    for (int i = threadIdx.x; i < n; i += 256) {
      const unsigned key = float_image(interp[lo + i]);
      if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned k = s_k;
      int bin = 255;
      for (; bin > 0; --bin) {
        if (hist[bin] >= k) break;
        k -= hist[bin];
      }
      s_prefix = prefix | ((unsigned)bin << shift);
      s_k = k;
    }
    mask |= 0xFFu << shift;
    __syncthreads();
  }
  float best = -__builtin_inff();
  bool any_nan = false;
  for (int i = threadIdx.x; i < n; i += 256) {
    const float x = ff[lo + i];
    any_nan |= x != x;
    best = x > best ? x : best;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const float o = __shfl_xor(best, off);
    best = o > best ? o : best;
  }
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    best = s_max[0];
    for (int w = 1; w < 4; ++w) best = s_max[w] > best ? s_max[w] : best;
    const float kth = image_float(s_prefix);
    const float last = sparse[lo + n - 1];
    if (bound_f64) {
      const double bound = __dadd_rn(__dmul_rn(a64, (double)last), __dmul_rn(b64, (double)best));
      active[q] = (double)kth < bound ? 1 : 0;
    } else {
      const float bound = __fadd_rn(__fmul_rn(a, last), __fmul_rn(b, best));
      active[q] = kth < bound ? 1 : 0;
    }
  }
  (void)any_nan;
}

// candidates of the next interval: cnt[q] = active ? |ranks [a, b) of q| : 0 (+ a zero sentinel)
__global__ void __launch_bounds__(256)
es_count_kernel(const int64_t* __restrict__ pair_off, int n_queries, const int32_t* __restrict__ active, int a, int b,
                int64_t* __restrict__ cnt) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q > n_queries) return;
  int64_t c = 0;
  if (q < n_queries && active[q]) {
    const int64_t len = pair_off[q + 1] - pair_off[q];
    const int64_t hi = len < b ? len : b, lo = len < a ? len : a;
    c = hi - lo;
  }
  cnt[q] = c;
}

// the interval's own CSR: ordinals, first-stage scores and the position of every pair in the full list
__global__ void __launch_bounds__(256)
es_fill_kernel(const int64_t* __restrict__ pair_off, const int32_t* __restrict__ pair_ids, const float* __restrict__ sparse,
               const int64_t* __restrict__ sub_off, int a, int32_t* __restrict__ sub_ids, float* __restrict__ sub_sparse,
               int32_t* __restrict__ sub_pos) {
  const int q = blockIdx.x;
  const int64_t dst = sub_off[q];
  const int64_t n = sub_off[q + 1] - dst;
  const int64_t src = pair_off[q] + a;
  for (int64_t j = threadIdx.x; j < n; j += blockDim.x) {
    sub_ids[dst + j] = pair_ids[src + j];
    sub_sparse[dst + j] = sparse[src + j];
    sub_pos[dst + j] = (int32_t)(src + j);
  }
}

__global__ void __launch_bounds__(256)
es_scatter_kernel(const float* __restrict__ sub_ff, const float* __restrict__ sub_interp, const int32_t* __restrict__ sub_pos,
                  int64_t n, float* __restrict__ ff, float* __restrict__ interp) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t p = sub_pos[i];
  ff[p] = sub_ff[i];
  interp[p] = sub_interp[i];
}

__global__ void __launch_bounds__(256)
fill_u32_kernel(uint32_t* __restrict__ dst, int64_t n, uint32_t value) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = value;
}

__global__ void __launch_bounds__(256)
interp_kernel(const float* __restrict__ ff, const float* __restrict__ sparse, int64_t n_pairs,
              float a, float b, float* __restrict__ interp) {
  // four candidates per thread (16-byte loads and stores; the buffers come from hipMalloc and 4 * thread is aligned)
  const int64_t p0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (p0 >= n_pairs) return;
  if (p0 + 4 <= n_pairs) {
    const float4 s = *reinterpret_cast<const float4*>(sparse + p0);
    const float4 d = *reinterpret_cast<const float4*>(ff + p0);
    float4 r;
    r.x = interp_one(a, b, s.x, d.x);
    r.y = interp_one(a, b, s.y, d.y);
    r.z = interp_one(a, b, s.z, d.z);
    r.w = interp_one(a, b, s.w, d.w);
    *reinterpret_cast<float4*>(interp + p0) = r;
    return;
  }
  for (int64_t p = p0; p < n_pairs; ++p) interp[p] = interp_one(a, b, sparse[p], ff[p]);
}

// owner-side merge of a row-sharded index: dense score of candidate p sits at recv[slot[p]]
__global__ void __launch_bounds__(256)
merge_kernel(const float* __restrict__ recv, const int32_t* __restrict__ slot, const float* __restrict__ sparse,
             int64_t n_pairs, float a, float b, float* __restrict__ ff, float* __restrict__ interp) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  const float dense = recv[slot[p]];
  ff[p] = dense;
  if (sparse != nullptr) interp[p] = interp_one(a, b, sparse[p], dense);
}

// Rows of an on-disk index pulled by the device itself: `chunks` are DEVICE addresses of page-locked chunk maps in host
// memory (ffx_host_pin + hipHostGetDevicePointer), so every load below is a read over PCIe and no host thread touches
// the rows.  Row rows[r] (chunk rows[r] / chunk_rows, row rows[r] % chunk_rows of it) lands in row r of `dst`.  A wave
// (G = 64, rows of 64 units and more) or a group of G lanes (a power of two) moves one row, up to four loads of V in flight
// per lane and row before the stores: ~55 GB/s of distinct 3072-byte rows (tools/pull_bench.py, tools/probes/host_pull_probe.hip;
// a page-locked hipMemcpy of one contiguous block: 57 GB/s on the same box).  Launched with 64 workgroups: more move no
// more bytes and their outstanding reads hold up every other kernel's loads (ffx_index_pull_rows_async).
template <typename V>
__global__ void __launch_bounds__(256)
pull_rows_kernel(const char* const* __restrict__ chunks, int64_t chunk_rows, const int64_t* __restrict__ rows, int64_t n_rows,
                 int row_units, int G, char* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = ((int64_t)gridDim.x * 256) >> 6;
  if (G == 64) {
    // A wave takes 64 consecutive rows at a time: every lane looks up ONE row's address (row number, chunk table, the
    // 64-bit division -- once per 64 rows, off the per-row path), then the rows are moved two at a time with the
    // addresses handed round by lane shuffles, so nothing but the reads over the link stands between two rows.
    for (int64_t base = wave * 64; base < n_rows; base += n_waves * 64) {
      const int cnt = (int)(n_rows - base < 64 ? n_rows - base : 64);
      unsigned long long mine = 0;
      if (lane < cnt) {
        const int64_t row = rows[base + lane];
        mine = (unsigned long long)(reinterpret_cast<uintptr_t>(chunks[row / chunk_rows]) + (size_t)(row % chunk_rows) * row_units * sizeof(V));
      }
      const unsigned lo = (unsigned)mine, hi = (unsigned)(mine >> 32);
      for (int i = 0; i < cnt; i += 2) {
        const bool two = i + 1 < cnt;
        const V* s0 = reinterpret_cast<const V*>(((unsigned long long)(unsigned)__shfl((int)hi, i) << 32) | (unsigned)__shfl((int)lo, i));
        const V* s1 = reinterpret_cast<const V*>(((unsigned long long)(unsigned)__shfl((int)hi, two ? i + 1 : i) << 32) | (unsigned)__shfl((int)lo, two ? i + 1 : i));
        V* d0 = reinterpret_cast<V*>(dst) + (size_t)(base + i) * row_units;
        V* d1 = d0 + row_units;
        for (int u0 = 0; u0 < row_units; u0 += 4 * 64) {
          V a[4], b[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int u = u0 + j * 64 + lane;
            if (u < row_units) a[j] = s0[u];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int u = u0 + j * 64 + lane;
            if (two && u < row_units) b[j] = s1[u];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int u = u0 + j * 64 + lane;
            if (u < row_units) d0[u] = a[j];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int u = u0 + j * 64 + lane;
            if (two && u < row_units) d1[u] = b[j];
          }
        }
      }
    }
    return;
  }
  // short rows (fewer than 64 units): a group of G lanes per row, several rows per wave
  const int gl = lane & (G - 1), per_wave = 64 / G;
  for (int64_t r = wave * per_wave + lane / G; r < n_rows; r += n_waves * per_wave) {
    const int64_t row = rows[r];
    const V* s = reinterpret_cast<const V*>(chunks[row / chunk_rows]) + (size_t)(row % chunk_rows) * row_units;
    V* d = reinterpret_cast<V*>(dst) + (size_t)r * row_units;
    for (int u0 = 0; u0 < row_units; u0 += 4 * G) {
      V v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int u = u0 + j * G + gl;
        if (u < row_units) v[j] = s[u];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int u = u0 + j * G + gl;
        if (u < row_units) d[u] = v[j];
      }
    }
  }
}

// ---- row copy-out for _get_vectors (index/memory.py:139-140) --------------------------------
template <typename W>
__global__ void __launch_bounds__(256)
gather_rows_kernel(const W* __restrict__ vectors, int d, const int32_t* __restrict__ rows,
                   int64_t n, W* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (r >= n) return;
  const W* src = vectors + (size_t)rows[r] * d;
  W* dst = out + (size_t)r * d;
  for (int c = threadIdx.x & 63; c < d; c += 64) dst[c] = src[c];
}

// ---- synthetic rows: same counter-based generator as oracle/ffx_oracle.c and synth.py ---------
__device__ __forceinline__ uint64_t mix64(uint64_t idx, uint64_t seed) {
  uint64_t z = idx + seed * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <typename T>
__global__ void __launch_bounds__(256)
fill_synthetic_kernel(T* __restrict__ dst, int64_t n_elems, uint64_t first_elem, uint64_t seed,
                      float scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_elems; e += stride) {
    const uint32_t m = (uint32_t)(mix64(first_elem + (uint64_t)e, seed) >> 40);
    const float u = __fsub_rn(__fmul_rn((float)m, 0x1p-23f), 1.0f);
    float x = __fmul_rn(u, scale);
    // keep the fp32 product as its own value: without this the compiler folds multiply + narrowing
    // into v_fma_mixlo_f16, and the half result must be round(round32(u*scale)) like NumPy's
    // float32 product followed by astype(float16)
    asm volatile("" : "+v"(x));
    dst[e] = (T)x;
  }
}

// synthetic product-quantizer codes: byte e = mix(e) mod Ks
__global__ void __launch_bounds__(256)
fill_codes_kernel(uint8_t* __restrict__ dst, int64_t n_elems, uint64_t first_elem, uint64_t seed, int ks) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_elems; e += stride)
    dst[e] = (uint8_t)((mix64(first_elem + (uint64_t)e, seed) >> 32) % (uint64_t)ks);
}

__global__ void __launch_bounds__(256)
iota_kernel(int32_t* __restrict__ dst, int64_t n, int32_t base) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = base + (int32_t)i;
}

// ---- per-query order, one workgroup per query, entirely in LDS -----------------------------------
// rocPRIM's segmented sort keeps queries of up to ~4 K candidates inside a workgroup, but sends longer
// ones (k = 5000 is just past that) through radix passes over global memory: 1.3 ms for 6980 x 5000.
// Here a query of up to BS * IPT candidates is loaded once (coalesced, through an LDS stage into the
// blocked arrangement a stable sort needs), ordered by rocprim::block_radix_sort in LDS -- keys are the
// order-preserving unsigned images of the scores (-0.0 counted as +0.0), values the positions, so
// "descending, exact ties by position" is the plain stable descending sort -- and stored once.
// Padding items carry image 0 behind every real item (stability), so they never reach the output.
// Where a candidate's sort key comes from: an array of keys, or -- the launch before the ordering folded into it -- made here:
// the dense score of candidate p (dense[p], or dense[slot[p]] at the owner of a sharded query, whose scores arrive grouped by
// the rank that sent them), interpolated with its first-stage score (ranking.py:319, two multiplications and one addition,
// each rounded), both stored on the way.
struct SortKeys {
  const float* keys = nullptr;    // plain form
  const float* dense = nullptr;   // made-here form ...
  const int32_t* slot = nullptr;  // ... read through the arrival slots
  const float* sparse = nullptr;  // NULL: the key is the dense score
  float a = 0.f, b = 0.f;
  int key_is_dense = 0;           // order by the dense score although first-stage scores are given (FFX_FLAG_SORT_FF)
  float* ff_out = nullptr;        // dense scores in first-stage order (the slot form)
  float* interp_out = nullptr;
};

__device__ inline float sort_key_of(const SortKeys& sk, int64_t p) {
  if (sk.keys) return sk.keys[p];
  const float d = sk.dense[sk.slot ? sk.slot[p] : p];
  if (sk.ff_out) sk.ff_out[p] = d;
  if (!sk.sparse) return d;
  const float mixed = __fadd_rn(__fmul_rn(sk.a, sk.sparse[p]), __fmul_rn(sk.b, d));
  sk.interp_out[p] = mixed;
  return sk.key_is_dense ? d : mixed;
}

template <unsigned BS, unsigned IPT>
__global__ void __launch_bounds__(BS)
block_sort_desc_kernel(SortKeys sk, const int64_t* __restrict__ pair_off, int32_t base,
                       int32_t* __restrict__ order) {
  using Sort = rocprim::block_radix_sort<uint32_t, BS, IPT, int32_t>;
  __shared__ typename Sort::storage_type storage;
  __shared__ uint32_t stage[BS * IPT];
  const int q = blockIdx.x;
  const int64_t lo = pair_off[q];
  const int n = (int)(pair_off[q + 1] - lo);
  if (n == 0) return;  // block-uniform
  for (int i = threadIdx.x; i < (int)(BS * IPT); i += BS) {
    uint32_t image = 0u;
    if (i < n) {
      uint32_t b = __float_as_uint(sort_key_of(sk, lo + i));
      if (b == 0x80000000u) b = 0u;
      image = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    }
    stage[i] = image;
  }
  __syncthreads();
  uint32_t k[IPT];
  int32_t v[IPT];
#pragma unroll
  for (unsigned i = 0; i < IPT; ++i) {
    const int at = (int)(threadIdx.x * IPT + i);
    k[i] = stage[at];
    v[i] = at;
  }
  __syncthreads();
  Sort().sort_desc(k, v, storage);
  __syncthreads();
#pragma unroll
  for (unsigned i = 0; i < IPT; ++i) stage[threadIdx.x * IPT + i] = (uint32_t)v[i];
  __syncthreads();
  const int32_t first = base + (int32_t)lo;
  for (int i = threadIdx.x; i < n; i += BS) order[lo + i] = first + (int32_t)stage[i];
}

// ---- per-query order of batches with very long queries ---------------------------------------
// The segmented sort gives every query to one workgroup, which is right for thousands of
// candidates and hopeless for millions.  Such batches are ordered by ONE device-wide stable radix
// sort instead, on 64-bit words  (query number) << 32 | ~(order-preserving image of the score):
// ascending words = queries in order, scores descending inside, exact ties by position
// (stability), -0.0 counted as +0.0.
__global__ void __launch_bounds__(256)
sort_words_kernel(const float* __restrict__ keys, const int64_t* __restrict__ pair_off, int n_queries,
                  double queries_per_pair, int64_t n_pairs, int32_t base, unsigned long long* __restrict__ words,
                  int32_t* __restrict__ positions) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pairs) return;
  uint32_t b = __float_as_uint(keys[p]);
  if (b == 0x80000000u) b = 0u;
  b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // larger score -> larger image
  const unsigned long long q = (unsigned long long)query_of_pair(pair_off, n_queries, p, queries_per_pair);
  words[p] = (q << 32) | (unsigned long long)(~b);
  positions[p] = base + (int32_t)p;
}

// ---- long rows as several short ones --------------------------------------------------------
// NumPy's pairwise sum of a row of d = S*b elements (S a power of two, b one of the row lengths the
// gather_dot kernels are specialised for) is the balanced pairwise sum of the S sums of its
// b-element segments.  So a row list over d-element rows becomes a list over the S*N "virtual"
// rows of b elements that the same memory holds (row r, segment s = virtual row r*S+s; likewise
// for the queries), the specialised kernel scores those, and the S partial sums are combined.
// Virtual rows are laid out in blocks of B rows per segment, so that a wave's tile has one segment
// (its query slice stays in registers); the tail of the last block repeats the last row.
__global__ void __launch_bounds__(256)
segment_desc_kernel(const int2* __restrict__ desc, int n_rows, int S, int B, int2* __restrict__ vdesc, long n_virtual) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_virtual) return;
  const long t = p / ((long)S * B);
  const int s = (int)((p / B) % S), i = (int)(p % B);
  long r = t * B + i;
  if (r > n_rows - 1) r = n_rows - 1;
  const int2 d = desc[r];
  vdesc[p] = make_int2(d.x * S + s, d.y * S + s);
}

__global__ void __launch_bounds__(256)
combine_segments_kernel(const float* __restrict__ rs_v, int n_rows, int S, int B, float* __restrict__ rs) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const long t = r / B;
  const int i = (int)(r % B);
  float v[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) v[s] = s < S ? rs_v[(t * S + s) * B + i] : 0.f;
#pragma unroll
  for (int st = 1; st < 8; st *= 2) {
    if (st < S) {
#pragma unroll
      for (int j = 0; j + st < 8; j += 2 * st) v[j] = __fadd_rn(v[j], v[j + st]);
    }
  }
  rs[r] = v[0];
}

// ---- placement probe (ffx_index_reseat) -------------------------------------------------------------------------------------
// Whole rows read at random, the way gather_dot reads them: a wave takes 8 rows, 4 at a time, every row in 16-byte coalesced
// non-temporal loads (each 128-B line requested once).  No arithmetic worth the name: the kernel is there to be timed -- a random
// gather over an allocation runs 3 - 6 % faster or slower with the physical memory the allocation was handed.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256)
placement_probe_kernel(const u32x4* __restrict__ base, long n_rows, int row_vec, long n_probe, unsigned long long seed, unsigned* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long first = wave * 8;
  if (first >= n_probe) return;
  unsigned acc = 0;
#pragma unroll 1
  for (int b = 0; b < 8; b += 4) {
    const u32x4* row[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      unsigned long long h = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(first + b + u + 1);  // splitmix64
      h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9ull;
      h = (h ^ (h >> 27)) * 0x94D049BB133111EBull;
      h ^= h >> 31;
      row[u] = base + (long)(h % (unsigned long long)n_rows) * row_vec;
    }
#pragma unroll 1
    for (int c0 = 0; c0 < row_vec; c0 += 256) {  // (a 3072-byte row: one round of 3 loads per row, 12 in flight)
      u32x4 v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = c0 + 64 * j + lane;
          v[u][j] = c < row_vec ? __builtin_nontemporal_load(row[u] + c) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc ^= v[u][j].x ^ v[u][j].y ^ v[u][j].z ^ v[u][j].w;
    }
  }
  sink[(wave * 64 + lane) & 65535] = acc;  // (what keeps the loads)
}

}  // namespace ffx
