// gather_dot.hip.h -- the hot kernels: gather index rows by id and take the inner product
// with the owning query's vector.  gfx950 (MI355X, CDNA4) only; wave = 64 lanes.
//
// Replaces src/fast_forward/index/base.py:301-303 of the reference
//     q_reps = query_vectors[q_no];  d_reps = vectors[id_idx];
//     ff_score = np.sum(q_reps * d_reps, axis=1)
// without materialising q_reps / d_reps: one wave streams a row from HBM straight into
// registers, multiplies by the query it already holds, and reduces across the wave.
//
// Work list: `desc[r] = {index row, query number}` for r in [0, R), rows of one query
// contiguous (built by ffx_expand_kernel).  Output: rs[r] = <query, row>.
//
// Rows are stored as T = float or _Float16 (the reference stores whatever dtype it is given,
// index/memory.py:80-83; float32 queries times float16 rows promote to float32 products in
// NumPy, so a half row is converted to fp32 -- exactly -- and everything after is the same).
//
// Two summation orders:
//   exact  NumPy's order (products rounded to fp32, 8 strided accumulators per leaf of
//          <=128 elements, leaves combined pairwise).  For d = 8*LEAVES*NPER the 64
//          partial sums of one row ARE the 64 lanes of a wave: lane (leaf l, accumulator j)
//          owns elements l*8*NPER + j + 8*i, i = 0..NPER-1, adds them in order of i, and the
//          combine tree is the xor-butterfly 1,2,4 (inside a leaf) 8,16,32 (across leaves).
//          Scores are bit-identical to the reference, hence so is the run ordering.
//   fast   coalesced loads, FMA chain, butterfly.  |error| ~ d * 2^-24 * |q||v|.
//
// Roofline: HBM.  Per row the kernel must move d*sizeof(T) B of row data + 8 B of descriptor in
// and 4 B out; 2*d flop.  <= 1 flop/B, so MFMA is irrelevant.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ffx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// ---- cross-lane helpers (DPP: no LDS traffic) ------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
// partner value for the xor-butterfly steps; every step is a plain commutative fp32 add, so
// both partners end with identical bits
__device__ __forceinline__ float xor1(float x) { return dpp_mov<0xB1>(x); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float xor2(float x) { return dpp_mov<0x4E>(x); }   // quad_perm [2,3,0,1]
// after xor1/xor2 the 4 lanes of a quad agree, so mirroring 8 (16) lanes pairs quad A with
// quad B (half-row A with half-row B) exactly as xor 4 (xor 8) would
__device__ __forceinline__ float xor4(float x) { return dpp_mov<0x141>(x); }  // row_half_mirror
__device__ __forceinline__ float xor8(float x) { return dpp_mov<0x140>(x); }  // row_mirror
__device__ __forceinline__ float xor16(float x) { return __shfl_xor(x, 16, 64); }
__device__ __forceinline__ float xor32(float x) { return __shfl_xor(x, 32, 64); }

// combine the 8*LEAVES partial sums of one row; all lanes of the row end with the total
template <int LEAVES>
__device__ __forceinline__ float butterfly(float x) {
  x = __fadd_rn(x, xor1(x));
  x = __fadd_rn(x, xor2(x));
  x = __fadd_rn(x, xor4(x));
  if constexpr (LEAVES >= 2) x = __fadd_rn(x, xor8(x));
  if constexpr (LEAVES >= 4) x = __fadd_rn(x, xor16(x));
  if constexpr (LEAVES >= 8) x = __fadd_rn(x, xor32(x));
  return x;
}

// ---- row loads ------------------------------------------------------------------------------
// read-once data, optionally marked non-temporal (nt) so it does not displace the query vectors
// and descriptors in L2.  Conversions half -> float are exact.
template <typename T, bool NT>
__device__ __forceinline__ float load1(const T* p) {
  if constexpr (NT) return (float)__builtin_nontemporal_load(p);
  else return (float)*p;
}

// 4 consecutive elements (16 B of float / 8 B of half), naturally aligned.  Loaded values stay
// in their storage type while they wait in registers (a half row in flight costs half the VGPRs)
// and are widened where they are used.
template <typename T> struct raw4 { typedef f32x4 type; };
template <> struct raw4<_Float16> { typedef f16x4 type; };

template <typename T, bool NT>
__device__ __forceinline__ typename raw4<T>::type load4raw(const T* p) {
  const typename raw4<T>::type* q = reinterpret_cast<const typename raw4<T>::type*>(p);
  if constexpr (NT) return __builtin_nontemporal_load(q);
  else return *q;
}

template <typename V>
__device__ __forceinline__ float4 widen4(V t) {
  return make_float4((float)t.x, (float)t.y, (float)t.z, (float)t.w);
}

template <typename T, bool NT>
__device__ __forceinline__ float4 load4(const T* p) { return widen4(load4raw<T, NT>(p)); }

constexpr int exact_default_passes(int leaves) {
  const int rpw = 64 / (8 * leaves);
  return rpw == 1 ? 16 : ((64 / rpw) < 16 ? (64 / rpw) : 16);
}

// ---- exact kernel, d = 8*LEAVES*NPER, LEAVES in {1,2,4,8}, NPER <= 16 ---------------------
// A wave owns TILE consecutive work rows.  LPR = 8*LEAVES lanes cooperate on one row, so a
// wave works on RPW = 64/LPR rows per pass and keeps U passes of loads in flight.  Each lane
// loads exactly the NPER elements it sums; the four load instructions that share a 128-B line
// merge in L1 (so no `nt` here: it would bypass L1 and quadruple the L2 traffic).
template <typename T, int LEAVES, int NPER, int U, int PASSES = exact_default_passes(LEAVES), bool NT = false>
__global__ void __launch_bounds__(256)
gather_dot_exact_kernel(const T* __restrict__ vectors, const float* __restrict__ queries,
                        const int2* __restrict__ desc, float* __restrict__ rs, int n_rows) {
  constexpr int LPR = 8 * LEAVES;
  constexpr int RPW = 64 / LPR;
  constexpr int D = LPR * NPER;
  constexpr int TILE = PASSES * RPW;
  static_assert(PASSES % U == 0, "PASSES must be a multiple of U");
  static_assert(TILE <= 64, "one descriptor per lane");

  const int lane = threadIdx.x & 63;
  // threadIdx.x >> 6 is wave-uniform; readfirstlane tells the compiler so (scalar addressing)
  const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long base = (long)wave * TILE;
  if (base >= n_rows) return;
  const int n_here = (n_rows - base) < TILE ? (int)(n_rows - base) : TILE;

  // one coalesced descriptor load per wave; lanes past the end repeat the last valid row so
  // every load below stays in bounds
  const int my_t = lane < n_here ? lane : n_here - 1;
  const int2 my = desc[base + my_t];

  const int sub = lane & (LPR - 1);
  const int slot = lane / LPR;                          // which row of the pass (0 if LPR == 64)
  const int lane_off = (sub >> 3) * (8 * NPER) + (sub & 7);

  if constexpr (RPW == 1) {
    float q[NPER];
    int cur_q = -1;
#pragma unroll
    for (int b = 0; b < PASSES; b += U) {
      if (b >= n_here) break;
      float v[U][NPER];
      int qn[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int t = (b + u) < n_here ? (b + u) : n_here - 1;
        const int rid = __builtin_amdgcn_readlane(my.x, t);
        qn[u] = __builtin_amdgcn_readlane(my.y, t);
        const T* rowp = vectors + (size_t)rid * D + lane_off;
#pragma unroll
        for (int i = 0; i < NPER; ++i) v[u][i] = load1<T, NT>(rowp + 8 * i);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (qn[u] != cur_q) {  // wave-uniform; rare (rows of a query are contiguous)
          cur_q = qn[u];
          const float* qp = queries + (size_t)cur_q * D + lane_off;
#pragma unroll
          for (int i = 0; i < NPER; ++i) q[i] = qp[8 * i];
        }
        float acc = __fmul_rn(q[0], v[u][0]);
#pragma unroll
        for (int i = 1; i < NPER; ++i) acc = __fadd_rn(acc, __fmul_rn(q[i], v[u][i]));
        acc = butterfly<LEAVES>(acc);
        if (lane == 0 && (b + u) < n_here) rs[base + b + u] = acc;
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < PASSES; b += U) {
      if (b * RPW >= n_here) break;
      float v[U][NPER], q[U][NPER];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int t = (b + u) * RPW + slot;
        t = t < n_here ? t : n_here - 1;
        const int rid = __shfl(my.x, t, 64);
        const int qn = __shfl(my.y, t, 64);
        const T* rowp = vectors + (size_t)rid * D + lane_off;
        const float* qp = queries + (size_t)qn * D + lane_off;
#pragma unroll
        for (int i = 0; i < NPER; ++i) v[u][i] = load1<T, false>(rowp + 8 * i);
#pragma unroll
        for (int i = 0; i < NPER; ++i) q[u][i] = qp[8 * i];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float acc = __fmul_rn(q[u][0], v[u][0]);
#pragma unroll
        for (int i = 1; i < NPER; ++i) acc = __fadd_rn(acc, __fmul_rn(q[u][i], v[u][i]));
        acc = butterfly<LEAVES>(acc);
        const int t = (b + u) * RPW + slot;
        if (sub == 0 && t < n_here) rs[base + t] = acc;
      }
    }
  }
}

// ---- exact kernel with full-line loads, d = 256*NCH with 8 leaves (768, 1024) -----------------
// Same arithmetic as gather_dot_exact_kernel (bit-identical), different data movement: rows come
// in with 16-byte coalesced, non-temporal loads (every 128-B line requested once, whole), the
// products are formed in that layout (a product does not care where it is computed), and a
// wave-private LDS tile turns them into NumPy's layout: lane (leaf l, accumulator j) reads back
// products l*LL + j + 8*i, i = 0..NPER-1, and adds them in order.  Leaves are padded from LL to
// LL+8 floats in LDS so that the 32 lanes of a ds_read_b32 group hit 32 different banks
// ((LL+8)*l + j mod 32 = 8*l + j for LL a multiple of 32).  LDS operations of one wave execute in
// order, so the read-after-write and the next row's write-after-read need no s_barrier.
// DERIVE (documents of several passages, Mode.MAXP / AVEP): there is no row list.  The wave works out the rows of its tile
// itself (get_indices, index/util.py:29-41 + the left merge of index/base.py:296-298, for TILE rows at a time): `tile_tab`
// names the candidate its first row belongs to and that candidate's query; TILE + 1 lanes read the row offsets of that
// candidate and the TILE after it (every candidate has at least one row, so the tile's rows lie among them), every lane finds
// its row's candidate by comparing against those offsets, and the row is that candidate's first row (`pair_first`: where the
// document's rows begin in doc_rows) plus the distance -- 4 bytes per candidate and 8 per tile instead of 8 per row written by
// an expand kernel and read back here (1.26 GB each way at 157 M rows).
// DERIVE = 2 (one row per candidate: PASSAGE, FIRSTP, documents of one passage): row t of the tile IS candidate base + t -- its
// row is `row_of[ordinal]` (or the ordinal itself when ordinals are rows), its query the one whose candidate list holds that
// position (found from the host's hint -- the query of every (1 << kQueryHintShift)-th candidate -- or, without one, from a
// proportional guess; then stepped forward per lane).  No expand launch and no row list here either: the kernel reads 4 bytes
// per candidate where it read 8.
constexpr int kQueryHintShift = 10;  // the hint names the query of candidates 0, 1024, 2048, ... (ffx_plan::kHintBlock in ffx.hip)
struct Derive {
  const int2* tile_tab = nullptr;       // per tile: {candidate of the tile's first row, query of that candidate}
  const int32_t* seg_off = nullptr;     // [n_pairs + 1] first row of every candidate in the (virtual) row list
  const int32_t* pair_first = nullptr;  // [n_pairs] position in doc_rows of every candidate's first row
  const int32_t* doc_rows = nullptr;    // nullptr: position == row (documents added whole, in order)
  const int64_t* pair_off = nullptr;    // [n_queries + 1]
  int n_pairs = 0;
  // DERIVE = 2
  const int32_t* pair_ids = nullptr;  // [n_pairs] ordinals
  const int32_t* row_of = nullptr;    // ordinal -> row; nullptr: ordinal == row
  int n_queries = 0;
  int n_index_rows = 0;
  double queries_per_pair = 0.0;
  const int32_t* q_hint = nullptr;    // query of candidate 0, 1024, 2048, ... and (last entry) the last query; nullptr: not given
};

// {row, query} of candidate base + my_t when every candidate has one row (DERIVE = 2)
__device__ __forceinline__ int2 derive_one_row(const Derive& dv, long base, int my_t) {
  const long p = base + my_t;
  const int o = dv.pair_ids[p];
  int rid = dv.row_of ? dv.row_of[o] : o;
  // (validated ordinals always give a row of the store; the clamp is for ffx_score's single-submission path, where the verdict
  // on a bad ordinal is read back only after the kernels behind this one have run)
  rid = (unsigned)rid < (unsigned)dv.n_index_rows ? rid : 0;
  // the query of the tile's first candidate (uniform), then forward: a tile rarely crosses a query boundary.  With the host's
  // hint (the query of every 1024th candidate) the search runs between two neighbouring entries: no step at all where
  // queries are longer than 1024 candidates, however ragged the lists; without it (lists made on the device) the
  // proportional guess, then a search over all queries
  int qn;
  if (dv.q_hint) {
    int lo = dv.q_hint[base >> kQueryHintShift], hi = dv.q_hint[(base >> kQueryHintShift) + 1];
    while (lo < hi) {  // the last query that begins at or before `base`
      const int mid = (lo + hi + 1) >> 1;
      if (dv.pair_off[mid] <= base) lo = mid; else hi = mid - 1;
    }
    qn = lo;
  } else {
    qn = (int)((double)base * dv.queries_per_pair);
    qn = qn < 0 ? 0 : (qn > dv.n_queries - 1 ? dv.n_queries - 1 : qn);
    if (!(dv.pair_off[qn] <= base && base < dv.pair_off[qn + 1])) {
      int lo = 0, hi = dv.n_queries;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (dv.pair_off[mid] <= base) lo = mid; else hi = mid;
      }
      qn = lo;
    }
  }
  long next_q = dv.pair_off[qn + 1];
  while (p >= next_q) {
    ++qn;
    next_q = dv.pair_off[qn + 1];
  }
  return make_int2(rid, qn);
}

// {row, query} of row base + my_t of a wave's tile: read from the row list (DERIVE = 0) or worked out here (see above);
// `wave` = the tile's number (DERIVE = 1: tile_tab was made for tiles of TILE rows)
template <int DERIVE, int TILE>
__device__ __forceinline__ int2 tile_row(const int2* __restrict__ desc, const Derive& dv, int wave, long base, int my_t, int lane) {
  if constexpr (DERIVE == 2) {
    return derive_one_row(dv, base, my_t);
  } else if constexpr (DERIVE == 1) {
    static_assert(TILE < 64, "TILE + 1 lanes read the row offsets");
    const int2 first = dv.tile_tab[wave];  // (uniform: one scalar load)
    const int p_mine = first.x + lane;
    const int so = lane <= TILE ? dv.seg_off[p_mine < dv.n_pairs ? p_mine : dv.n_pairs] : 0x7fffffff;
    const int pf = dv.pair_first[p_mine < dv.n_pairs ? p_mine : dv.n_pairs - 1];
    long next_q = dv.pair_off[first.y + 1];
    const int r = (int)base + my_t;
    int j = 0;  // how many of the TILE candidates after the first begin at or before my row
#pragma unroll
    for (int l = 1; l <= TILE; ++l) j += __builtin_amdgcn_readlane(so, l) <= r ? 1 : 0;
    int rid = __shfl(pf, j) + (r - __shfl(so, j));
    if (dv.doc_rows) rid = dv.doc_rows[rid];
    int qn = first.y;
    while (first.x + j >= next_q) {  // my candidate lies in a later query (queries without candidates are stepped over)
      ++qn;
      next_q = dv.pair_off[qn + 1];
    }
    return make_int2(rid, qn);
  } else {
    return desc[base + my_t];
  }
}

template <typename T, int NCH, int U, int TILE, bool NT, int DERIVE = 0>
__global__ void __launch_bounds__(256)
gather_dot_exact_lds_kernel(const T* __restrict__ vectors, const float* __restrict__ queries,
                            const int2* __restrict__ desc, float* __restrict__ rs, int n_rows, const Derive dv) {
  constexpr int D = 256 * NCH;
  constexpr int LL = D / 8;        // leaf length (96 or 128)
  constexpr int NPER = LL / 8;     // adds per accumulator
  constexpr int LPAD = LL + 8;     // padded leaf stride in LDS
  static_assert(LL % 32 == 0 && LL <= 128, "8 leaves of at most 128 elements");
  static_assert(TILE % U == 0 && TILE <= 64, "bad tile");
  __shared__ __attribute__((aligned(16))) float tile[4][8 * LPAD];

  const int lane = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = blockIdx.x * (blockDim.x >> 6) + wave_in_block;
  const long base = (long)wave * TILE;
  if (base >= n_rows) return;
  const int n_here = (n_rows - base) < TILE ? (int)(n_rows - base) : TILE;
  const int my_t = lane < n_here ? lane : n_here - 1;
  const int2 my = tile_row<DERIVE, TILE>(desc, dv, wave, base, my_t, lane);

  float* buf = tile[wave_in_block];
  // where this lane's NCH float4 of products go, and where its NPER summands come from
  int woff[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = 4 * lane + 256 * c;
    woff[c] = (e / LL) * LPAD + (e % LL);
  }
  const float* rbuf = buf + (lane >> 3) * LPAD + (lane & 7);

  float4 q[NCH];
  int cur_q = -1;
#pragma unroll 1
  for (int b = 0; b < TILE; b += U) {
    if (b >= n_here) break;
    typename raw4<T>::type v[U][NCH];
    int qn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = (b + u) < n_here ? (b + u) : n_here - 1;
      const int rid = __builtin_amdgcn_readlane(my.x, t);
      qn[u] = __builtin_amdgcn_readlane(my.y, t);
      const T* rowp = vectors + (size_t)rid * D + 4 * lane;
#pragma unroll
      for (int c = 0; c < NCH; ++c) v[u][c] = load4raw<T, NT>(rowp + 256 * c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (qn[u] != cur_q) {
        cur_q = qn[u];
        const float4* qp = reinterpret_cast<const float4*>(queries + (size_t)cur_q * D) + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) q[c] = qp[64 * c];
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float4 w = widen4(v[u][c]);
        float4 p;
        p.x = __fmul_rn(q[c].x, w.x);
        p.y = __fmul_rn(q[c].y, w.y);
        p.z = __fmul_rn(q[c].z, w.z);
        p.w = __fmul_rn(q[c].w, w.w);
        *reinterpret_cast<float4*>(buf + woff[c]) = p;
      }
      __builtin_amdgcn_wave_barrier();
      float acc = rbuf[0];
#pragma unroll
      for (int i = 1; i < NPER; ++i) acc = __fadd_rn(acc, rbuf[8 * i]);
      __builtin_amdgcn_wave_barrier();
      acc = butterfly<8>(acc);
      if (lane == 0 && (b + u) < n_here) rs[base + b + u] = acc;
    }
  }
}

#ifdef FFX_EXPERIMENTS
// ---- experiment (FFX_DEBUG_VARIANT=gather=921 / 922 / 923, tools/kernel_ab.py): the same kernel as a PERSISTENT grid --------
// gridDim.x workgroups stay for the whole launch and every wave walks tiles of the row list: CHUNKED = false: tile w, w + W,
// w + 2 W ... (W waves in all); CHUNKED = true: a contiguous share of the tiles (consecutive rows: the query vector stays in
// registers for ~k / TILE tiles).  Same arithmetic, same bits.  Question put to it: does the turnover of a million short-lived
// workgroups (a workgroup's LDS and wave slots come free only when its last wave is done) cost bandwidth?
template <typename T, int NCH, int U, int TILE, bool NT, bool CHUNKED>
__global__ void __launch_bounds__(256)
gather_dot_exact_lds_persistent_kernel(const T* __restrict__ vectors, const float* __restrict__ queries,
                                       const int2* __restrict__ desc, float* __restrict__ rs, int n_rows) {
  constexpr int D = 256 * NCH;
  constexpr int LL = D / 8;
  constexpr int NPER = LL / 8;
  constexpr int LPAD = LL + 8;
  static_assert(LL % 32 == 0 && LL <= 128, "8 leaves of at most 128 elements");
  static_assert(TILE % U == 0 && TILE <= 64, "bad tile");
  __shared__ __attribute__((aligned(16))) float tile[4][8 * LPAD];

  const int lane = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + wave_in_block;
  const long n_waves = (long)gridDim.x * (blockDim.x >> 6);
  const long n_tiles = ((long)n_rows + TILE - 1) / TILE;
  const long per_wave = (n_tiles + n_waves - 1) / n_waves;
  long t_first = CHUNKED ? wave * per_wave : wave;
  const long t_end = CHUNKED ? (t_first + per_wave < n_tiles ? t_first + per_wave : n_tiles) : n_tiles;
  const long t_step = CHUNKED ? 1 : n_waves;

  float* buf = tile[wave_in_block];
  int woff[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = 4 * lane + 256 * c;
    woff[c] = (e / LL) * LPAD + (e % LL);
  }
  const float* rbuf = buf + (lane >> 3) * LPAD + (lane & 7);
  float4 q[NCH];
  int cur_q = -1;
#pragma unroll 1
  for (long t = t_first; t < t_end; t += t_step) {
    const long base = t * TILE;
    const int n_here = (n_rows - base) < TILE ? (int)(n_rows - base) : TILE;
    const int my_t = lane < n_here ? lane : n_here - 1;
    const int2 my = desc[base + my_t];
#pragma unroll 1
    for (int b = 0; b < TILE; b += U) {
      if (b >= n_here) break;
      typename raw4<T>::type v[U][NCH];
      int qn[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int tt = (b + u) < n_here ? (b + u) : n_here - 1;
        const int rid = __builtin_amdgcn_readlane(my.x, tt);
        qn[u] = __builtin_amdgcn_readlane(my.y, tt);
        const T* rowp = vectors + (size_t)rid * D + 4 * lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[u][c] = load4raw<T, NT>(rowp + 256 * c);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (qn[u] != cur_q) {
          cur_q = qn[u];
          const float4* qp = reinterpret_cast<const float4*>(queries + (size_t)cur_q * D) + lane;
#pragma unroll
          for (int c = 0; c < NCH; ++c) q[c] = qp[64 * c];
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          const float4 w = widen4(v[u][c]);
          float4 p;
          p.x = __fmul_rn(q[c].x, w.x);
          p.y = __fmul_rn(q[c].y, w.y);
          p.z = __fmul_rn(q[c].z, w.z);
          p.w = __fmul_rn(q[c].w, w.w);
          *reinterpret_cast<float4*>(buf + woff[c]) = p;
        }
        __builtin_amdgcn_wave_barrier();
        float acc = rbuf[0];
#pragma unroll
        for (int i = 1; i < NPER; ++i) acc = __fadd_rn(acc, rbuf[8 * i]);
        __builtin_amdgcn_wave_barrier();
        acc = butterfly<8>(acc);
        if (lane == 0 && (b + u) < n_here) rs[base + b + u] = acc;
      }
    }
  }
}

#endif  // FFX_EXPERIMENTS

// ---- the same for smaller rows: d = 8*LEAVES*NPER with LEAVES in {1,2,4}, NPER % 4 == 0 ----------
// (d = 512, 384, 256, 128, 64, ...).  NumPy's tree for such a row has 8*LEAVES partial sums, so
// RPP = 8/LEAVES rows fill a wave.  A pass takes RPP consecutive work rows and treats them as ONE
// row of 64*NPER floats with 8 leaves of LL = 8*NPER elements: element e of the concatenation lives
// in row e / D at offset e % D, leaf e / LL -- so the LDS transpose is the 8-leaf one above, the
// only differences being per-lane row/query addresses and a shorter butterfly (a row's partial
// sums occupy 8*LEAVES adjacent lanes; lane 8*LEAVES*s ends with the score of the pass's row s).
template <typename T, int LEAVES, int NPER, int U, int TILE, bool NT, int DERIVE = 0>
__global__ void __launch_bounds__(256)
gather_dot_exact_lds_multi_kernel(const T* __restrict__ vectors, const float* __restrict__ queries,
                                  const int2* __restrict__ desc, float* __restrict__ rs, int n_rows, const Derive dv) {
  constexpr int LPR = 8 * LEAVES;     // lanes per row in the summation phase
  constexpr int RPP = 64 / LPR;       // rows per pass
  constexpr int D = LPR * NPER;
  constexpr int LL = 8 * NPER;        // leaf length
  constexpr int LPAD = LL + 8;
  constexpr int NCH = NPER / 4;       // float4 groups per lane per pass
  static_assert(LEAVES == 1 || LEAVES == 2 || LEAVES == 4, "use gather_dot_exact_lds_kernel for 8 leaves");
  static_assert(NPER % 4 == 0 && LL % 32 == 0 && LL <= 128, "leaf length must be a multiple of 32");
  static_assert(TILE % (U * RPP) == 0 && TILE <= 64, "bad tile");
  __shared__ __attribute__((aligned(16))) float tile[4][8 * LPAD];

  const int lane = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = blockIdx.x * (blockDim.x >> 6) + wave_in_block;
  const long base = (long)wave * TILE;
  if (base >= n_rows) return;
  const int n_here = (n_rows - base) < TILE ? (int)(n_rows - base) : TILE;
  const int my_t = lane < n_here ? lane : n_here - 1;
  static_assert(DERIVE != 1, "several rows per candidate: the 8-leaf kernels only");
  const int2 my = tile_row<DERIVE, TILE>(desc, dv, wave, base, my_t, lane);

  float* buf = tile[wave_in_block];
  int woff[NCH], slot[NCH], roff[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = 4 * lane + 256 * c;
    woff[c] = (e / LL) * LPAD + (e % LL);
    slot[c] = e / D;  // which row of the pass these 4 elements belong to
    roff[c] = e % D;  // and where in that row
  }
  const float* rbuf = buf + (lane >> 3) * LPAD + (lane & 7);
  const int out_slot = lane / LPR;

  float4 q[NCH];
  int cur_q[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) cur_q[c] = -1;
#pragma unroll 1
  for (int b = 0; b < TILE; b += U * RPP) {
    if (b >= n_here) break;
    typename raw4<T>::type v[U][NCH];
    int qn[U][NCH];
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        int t = b + u * RPP + slot[c];
        t = t < n_here ? t : n_here - 1;
        const int rid = __shfl(my.x, t, 64);
        qn[u][c] = __shfl(my.y, t, 64);
        v[u][c] = load4raw<T, NT>(vectors + (size_t)rid * D + roff[c]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        if (__ballot(qn[u][c] != cur_q[c]) != 0ull) {  // wave-uniform: reload only when some lane's query changed
          cur_q[c] = qn[u][c];
          q[c] = *reinterpret_cast<const float4*>(queries + (size_t)cur_q[c] * D + roff[c]);
        }
        const float4 w = widen4(v[u][c]);
        float4 p;
        p.x = __fmul_rn(q[c].x, w.x);
        p.y = __fmul_rn(q[c].y, w.y);
        p.z = __fmul_rn(q[c].z, w.z);
        p.w = __fmul_rn(q[c].w, w.w);
        *reinterpret_cast<float4*>(buf + woff[c]) = p;
      }
      __builtin_amdgcn_wave_barrier();
      float acc = rbuf[0];
#pragma unroll
      for (int i = 1; i < NPER; ++i) acc = __fadd_rn(acc, rbuf[8 * i]);
      __builtin_amdgcn_wave_barrier();
      acc = butterfly<LEAVES>(acc);
      const int t = b + u * RPP + out_slot;
      if ((lane & (LPR - 1)) == 0 && t < n_here) rs[base + t] = acc;
    }
  }
}

// ---- fast kernel, d = 256*NCH: lane holds NCH float4 per row (16-B coalesced loads) --------
// DERIVE: as in gather_dot_exact_lds_kernel (no row list; 1 = several rows per candidate, 2 = one)
template <typename T, int NCH, int U, int TILE = 16, bool NT = false, int DERIVE = 0>
__global__ void __launch_bounds__(256)
gather_dot_fast_kernel(const T* __restrict__ vectors, const float* __restrict__ queries,
                       const int2* __restrict__ desc, float* __restrict__ rs, int n_rows, const Derive dv) {
  constexpr int D = 256 * NCH;
  static_assert(TILE % U == 0, "TILE must be a multiple of U");
  const int lane = threadIdx.x & 63;
  // threadIdx.x >> 6 is wave-uniform; readfirstlane tells the compiler so (scalar addressing)
  const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long base = (long)wave * TILE;
  if (base >= n_rows) return;
  const int n_here = (n_rows - base) < TILE ? (int)(n_rows - base) : TILE;
  const int my_t = lane < n_here ? lane : n_here - 1;
  const int2 my = tile_row<DERIVE, TILE>(desc, dv, wave, base, my_t, lane);

  float4 q[NCH];
  int cur_q = -1;
#pragma unroll
  for (int b = 0; b < TILE; b += U) {
    if (b >= n_here) break;
    typename raw4<T>::type v[U][NCH];
    int qn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = (b + u) < n_here ? (b + u) : n_here - 1;
      const int rid = __builtin_amdgcn_readlane(my.x, t);
      qn[u] = __builtin_amdgcn_readlane(my.y, t);
      const T* rowp = vectors + (size_t)rid * D + 4 * lane;
#pragma unroll
      for (int c = 0; c < NCH; ++c) v[u][c] = load4raw<T, NT>(rowp + 256 * c);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (qn[u] != cur_q) {
        cur_q = qn[u];
        const float4* qp = reinterpret_cast<const float4*>(queries + (size_t)cur_q * D) + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) q[c] = qp[64 * c];
      }
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const float4 w = widen4(v[u][c]);
        acc = fmaf(q[c].x, w.x, acc);
        acc = fmaf(q[c].y, w.y, acc);
        acc = fmaf(q[c].z, w.z, acc);
        acc = fmaf(q[c].w, w.w, acc);
      }
      acc = butterfly<8>(acc);
      if (lane == 0 && (b + u) < n_here) rs[base + b + u] = acc;
    }
  }
}

// ---- fast kernel, any d % 4 == 0 (runtime chunk loop) ----------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
gather_dot_fast_generic_kernel(const T* __restrict__ vectors, const float* __restrict__ queries,
                               const int2* __restrict__ desc, float* __restrict__ rs, int n_rows,
                               int d) {
  const int lane = threadIdx.x & 63;
  const long r = (long)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (r >= n_rows) return;
  const int2 de = desc[r];
  const T* rowp = vectors + (size_t)de.x * d;
  const float4* qp = reinterpret_cast<const float4*>(queries + (size_t)de.y * d);
  float acc = 0.f;
  for (int c = lane; c < (d >> 2); c += 64) {
    const float4 a = qp[c], b = load4<T, false>(rowp + 4 * c);
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
  }
  acc = butterfly<8>(acc);
  if (lane == 0) rs[r] = acc;
}

// ---- exact kernel, any d: leaves of NumPy's recursion handed to 8-lane groups -------------
// leaf_start[0..n_leaves] are the leaf boundaries in in-order (host: numpy_leaf_plan).
// One wave (= one block) per row; leaf sums go through LDS, lane 0 replays the recursion from
// the host-made list of steps (ffx.hip: numpy_combine_plan).

// ---- exact kernel for ANY row length with at most 8 leaves (8 <= d <= 1024) -----------------------
// The strided form of gather_dot_exact_kernel with the leaf table at run time.  NumPy cuts a row of
// d elements into leaves of <= 128 (numpy_leaf_plan on the host); inside a leaf, accumulator j sums
// elements s + j + 8*i in order of i, the 8 accumulators are tree-combined, the len % 8 trailing
// elements are then added one by one, and the leaf sums are added in the order of the recursion
// (combine_steps: "sum[a] += sum[b]").  LG = leaves per row rounded up to a power of two, so
// 8*LG lanes own a row and a wave takes 64 / (8*LG) rows per pass: lane (leaf l, accumulator j)
// loads exactly the <= 16 elements it sums, each 128-B line is shared by the four load
// instructions that follow one another and merges in L1 (the round-1 "strided dword" form: 90 % of
// the full-line + LDS-transpose kernel at d = 768), and nothing is staged or walked serially --
// the staged generic kernel below ran at 1.2 - 3.5 TB/s because one wave walked its leaves out of
// LDS element by element and lane 0 combined the leaf sums alone.
template <typename T, int LG, int U, bool TAILS>
__global__ void __launch_bounds__(256)
gather_dot_exact_anyd_kernel(const T* __restrict__ vectors, const float* __restrict__ queries,
                             const int2* __restrict__ desc, float* __restrict__ rs, int n_rows, int stride, int col0,
                             const int* __restrict__ leaf_start, int n_leaves, const int* __restrict__ combine_steps) {
  // stride: elements from one row (and one query vector) to the next; col0: first element of the SEGMENT of the row this
  // launch sums (0 and the whole row for d <= 1024; a longer row is cut along NumPy's recursion into segments of at
  // most 8 leaves each, one launch per segment: ffx.hip, long_row_plan)
  constexpr int LPR = 8 * LG;
  constexpr int RPW = 64 / LPR;
  constexpr int PASSES = exact_default_passes(LG);
  constexpr int TILE = PASSES * RPW;
  constexpr int NMAX = 16;  // a leaf has at most 128 elements
  static_assert(PASSES % U == 0 && TILE <= 64, "bad tiling");

  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long base = (long)wave * TILE;
  if (base >= n_rows) return;
  const int n_here = (n_rows - base) < TILE ? (int)(n_rows - base) : TILE;
  const int my_t = lane < n_here ? lane : n_here - 1;
  const int2 my = desc[base + my_t];

  const int sub = lane & (LPR - 1);
  const int slot = lane / LPR;
  const int leaf = sub >> 3, j = sub & 7;
  const bool live = leaf < n_leaves;
  const int s = live ? leaf_start[leaf] : 0;
  const int len = live ? leaf_start[leaf + 1] - s : 0;
  const int cnt = len >> 3;            // strided elements of this lane
  const int tail = len & 7;            // trailing elements of this leaf
  const int lane_off = s + j;
  const int tail_off = s + 8 * cnt + j;  // lane j holds trailing element j (j < tail)
  const int group0 = lane & ~7;          // first lane of this leaf's 8-lane group
  const int row0 = lane & ~(LPR - 1);    // first lane of this row's lanes

  // the query's elements this lane multiplies with live in registers and are reloaded only when the
  // query changes (rows of a query are contiguous, so nearly every pass serves one query)
  float q[NMAX], tq = 0.f;
  int cur_q = -1;
#pragma unroll 1
  for (int b = 0; b < PASSES; b += U) {
    if (b * RPW >= n_here) break;
    float v[U][NMAX], tv[U];
    int qn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int t = (b + u) * RPW + slot;
      t = t < n_here ? t : n_here - 1;
      const int rid = __shfl(my.x, t, 64);
      qn[u] = __shfl(my.y, t, 64);
      const T* rowp = vectors + (size_t)rid * stride + col0;
#pragma unroll
      for (int i = 0; i < NMAX; ++i) v[u][i] = i < cnt ? load1<T, false>(rowp + lane_off + 8 * i) : 0.f;
      if constexpr (TAILS) tv[u] = j < tail ? load1<T, false>(rowp + tail_off) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q0 = __builtin_amdgcn_readfirstlane(qn[u]);
      const bool one_query = __all(qn[u] == q0);  // wave-uniform
      if (!one_query || q0 != cur_q) {
        const float* qp = queries + (size_t)qn[u] * stride + col0;  // L2-resident
#pragma unroll
        for (int i = 0; i < NMAX; ++i) q[i] = i < cnt ? qp[lane_off + 8 * i] : 0.f;
        if constexpr (TAILS) tq = j < tail ? qp[tail_off] : 0.f;
        cur_q = one_query ? q0 : -1;
      }
      float acc = 0.f;
      if (cnt > 0) acc = __fmul_rn(q[0], v[u][0]);
#pragma unroll
      for (int i = 1; i < NMAX; ++i)
        if (i < cnt) acc = __fadd_rn(acc, __fmul_rn(q[i], v[u][i]));
      acc = __fadd_rn(acc, xor1(acc));
      acc = __fadd_rn(acc, xor2(acc));
      acc = __fadd_rn(acc, xor4(acc));
      if constexpr (TAILS) {
        const float tp = j < tail ? __fmul_rn(tq, tv[u]) : 0.f;
#pragma unroll
        for (int t = 0; t < 7; ++t) {  // res += a[lim + t], one by one, on every lane of the group alike
          const float x = __shfl(tp, group0 + t, 64);
          if (t < tail) acc = __fadd_rn(acc, x);
        }
      }
      // leaf sums -> row sum in the order of NumPy's recursion
      for (int t = 0; t < n_leaves - 1; ++t) {
        const int a = combine_steps[2 * t], bb = combine_steps[2 * t + 1];  // wave-uniform
        const float x = __shfl(acc, row0 + 8 * bb, 64);
        if (leaf == a) acc = __fadd_rn(acc, x);
      }
      const int t = (b + u) * RPW + slot;
      if (sub == 0 && t < n_here) rs[base + t] = acc;
    }
  }
}

// row accessors for the generic kernels: element i of the (decoded) row as fp32
template <typename T>
struct PlainRow {
  const T* p;
  __device__ __forceinline__ float at(int i) const { return (float)p[i]; }
};
// product-quantized row: element i = codeword[m][code_m][i % Ds], m = i / Ds
// (Quantizer.decode of the reference, index/base.py:292-293: concatenated codewords)
struct PqRow {
  const uint8_t* codes;   // M code bytes of this row
  const float* codebook;  // [M][Ks][Ds]
  int Ks, Ds;
  __device__ __forceinline__ float at(int i) const {
    const int m = i / Ds;
    const int code = codes[m] < Ks ? codes[m] : Ks - 1;  // never past the codebook (uploads reject such codes)
    return codebook[((size_t)m * Ks + code) * Ds + (i - m * Ds)];
  }
};

template <typename Row>
__device__ __forceinline__ void exact_generic_row(const Row row, const float* __restrict__ q, int d,
                                                  const int* __restrict__ leaf_start, int n_leaves,
                                                  const int* __restrict__ combine_steps, float* leaf_sum, float* out) {
  const int lane = threadIdx.x;
  const int g = lane >> 3, j = lane & 7;
  for (int leaf = g; leaf < n_leaves; leaf += 8) {
    const int s = leaf_start[leaf];
    const int len = leaf_start[leaf + 1] - s;
    float res;
    if (len < 8) {
      res = -0.0f;
      for (int i = 0; i < len; ++i) res = __fadd_rn(res, __fmul_rn(q[s + i], row.at(s + i)));
    } else {
      const int lim = len - (len % 8);
      float acc = __fmul_rn(q[s + j], row.at(s + j));
      for (int i = 8; i < lim; i += 8) acc = __fadd_rn(acc, __fmul_rn(q[s + i + j], row.at(s + i + j)));
      acc = __fadd_rn(acc, xor1(acc));
      acc = __fadd_rn(acc, xor2(acc));
      acc = __fadd_rn(acc, xor4(acc));
      res = acc;
      for (int i = lim; i < len; ++i) res = __fadd_rn(res, __fmul_rn(q[s + i], row.at(s + i)));
    }
    if (j == 0) leaf_sum[leaf] = res;
  }
  __syncthreads();
  if (lane == 0) {
    // NumPy's recursion over the leaf sums, unrolled on the host into "sum[a] += sum[b]" steps
    // (a recursive device function kept its frames in scratch memory and dominated the kernel)
    for (int t = 0; t < n_leaves - 1; ++t) {
      const int a = combine_steps[2 * t], b = combine_steps[2 * t + 1];
      leaf_sum[a] = __fadd_rn(leaf_sum[a], leaf_sum[b]);
    }
    *out = leaf_sum[0];
  }
}

template <typename T>
__global__ void __launch_bounds__(64)
gather_dot_exact_generic_kernel(const T* __restrict__ vectors, const float* __restrict__ queries,
                                const int2* __restrict__ desc, float* __restrict__ rs, int n_rows,
                                int d, const int* __restrict__ leaf_start, int n_leaves,
                                const int* __restrict__ combine_steps) {
  extern __shared__ float leaf_sum[];
  const long r = blockIdx.x;
  if (r >= n_rows) return;  // whole block leaves together
  const int2 de = desc[r];
  exact_generic_row(PlainRow<T>{vectors + (size_t)de.x * d}, queries + (size_t)de.y * d, d, leaf_start, n_leaves,
                    combine_steps, leaf_sum, rs + r);
}

// The same arithmetic with the rows staged in LDS first: a wave fetches its rows (and, when it
// changes, the query) with independent, coalesced loads -- the loop above waits for memory once per
// element of a leaf -- and then walks the leaves out of LDS.  Short rows have few leaves, so a
// wave takes RP = 8 / LG rows per pass (LG = leaves per row rounded up to a power of two, at most
// 8): 8-lane group g works on row g / LG, leaves g % LG, g % LG + LG, ...  Each wave owns its slice
// of LDS (RP rows | RP queries | RP x leaf sums) and a run of `rows_per_wave` consecutive rows of
// the work list; LDS operations of one wave execute in order, so the wave barrier is only a
// scheduling fence.
template <typename T>
__global__ void __launch_bounds__(256)
gather_dot_exact_generic_staged_kernel(const T* __restrict__ vectors, const float* __restrict__ queries,
                                       const int2* __restrict__ desc, float* __restrict__ rs, int n_rows, int d,
                                       const int* __restrict__ leaf_start, int n_leaves,
                                       const int* __restrict__ combine_steps, int dp, int lp, int rows_per_wave, int RP) {
  extern __shared__ float staged[];
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* row_s = staged + (size_t)w * RP * (2 * dp + lp);
  float* q_s = row_s + RP * dp;
  float* leaf_s = q_s + RP * dp;
  const long wave = (long)blockIdx.x * (blockDim.x >> 6) + w;
  long r = wave * rows_per_wave;
  const long r_end = r + rows_per_wave < (long)n_rows ? r + rows_per_wave : (long)n_rows;
  const int LG = 8 / RP;
  const int g = lane >> 3, j = lane & 7;
  const int slot = g / LG, leaf0 = g % LG;
  int cur_q = -1;  // the query held in slot 0 of q_s, if all rows of the last pass shared one
  for (; r < r_end; r += RP) {
    const int n_now = (int)(r_end - r < RP ? r_end - r : RP);
    int q_first = 0;
    bool same = true;
    for (int t = 0; t < n_now; ++t) {  // wave-uniform
      const int2 de = desc[r + t];
      const int row = __builtin_amdgcn_readfirstlane(de.x), q = __builtin_amdgcn_readfirstlane(de.y);
      if (t == 0) q_first = q;
      same = same && q == q_first;
      const T* src = vectors + (size_t)row * d;
      for (int i = lane; i < d; i += 64) row_s[t * dp + i] = (float)src[i];
    }
    if (same) {
      if (q_first != cur_q) {
        const float* qs = queries + (size_t)q_first * d;
        for (int i = lane; i < d; i += 64) q_s[i] = qs[i];
        cur_q = q_first;
      }
    } else {  // a query boundary inside the pass: every row brings its own query
      for (int t = 0; t < n_now; ++t) {
        const float* qs = queries + (size_t)__builtin_amdgcn_readfirstlane(desc[r + t].y) * d;
        for (int i = lane; i < d; i += 64) q_s[t * dp + i] = qs[i];
      }
      cur_q = -1;
    }
    __builtin_amdgcn_wave_barrier();
    if (slot < n_now) {
      const float* rowp = row_s + slot * dp;
      const float* qp = q_s + (same ? 0 : slot * dp);
      for (int leaf = leaf0; leaf < n_leaves; leaf += LG) {
        const int s0 = leaf_start[leaf];
        const int len = leaf_start[leaf + 1] - s0;
        float res;
        if (len < 8) {
          res = -0.0f;
          for (int i = 0; i < len; ++i) res = __fadd_rn(res, __fmul_rn(qp[s0 + i], rowp[s0 + i]));
        } else {
          const int lim = len - (len % 8);
          float acc = __fmul_rn(qp[s0 + j], rowp[s0 + j]);
          for (int i = 8; i < lim; i += 8) acc = __fadd_rn(acc, __fmul_rn(qp[s0 + i + j], rowp[s0 + i + j]));
          acc = __fadd_rn(acc, xor1(acc));
          acc = __fadd_rn(acc, xor2(acc));
          acc = __fadd_rn(acc, xor4(acc));
          res = acc;
          for (int i = lim; i < len; ++i) res = __fadd_rn(res, __fmul_rn(qp[s0 + i], rowp[s0 + i]));
        }
        if (j == 0) leaf_s[slot * lp + leaf] = res;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (j == 0 && leaf0 == 0 && slot < n_now) {
      // NumPy's recursion over the leaf sums, unrolled on the host into "sum[a] += sum[b]" steps
      float* sums = leaf_s + slot * lp;
      for (int t = 0; t < n_leaves - 1; ++t) {
        const int a = combine_steps[2 * t], b = combine_steps[2 * t + 1];
        sums[a] = __fadd_rn(sums[a], sums[b]);
      }
      rs[r + slot] = sums[0];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- product-quantized rows --------------------------------------------------------------------
// The index stores M code bytes per row; the codebook [M][Ks][Ds] fp32 (768 KB at M=8, Ks=256,
// Ds=96) stays resident in L2.  "decode-then-dot": the row the reference would build with
// quantizer.decode (index/base.py:292-293) is never materialised; its elements are fetched from
// the codebook as they are needed and scored with the same arithmetic as a stored fp32 row, so the
// scores are bit-identical to decode + np.sum(q * d).  HBM traffic per row drops from d*4 to M
// bytes; the kernel is bound by L2 -> CU codeword traffic (d*4 bytes per row) and the LDS transpose.
__global__ void __launch_bounds__(64)
gather_dot_pq_exact_generic_kernel(const uint8_t* __restrict__ codes, int M, const float* __restrict__ codebook,
                                   int Ks, int Ds, const float* __restrict__ queries,
                                   const int2* __restrict__ desc, float* __restrict__ rs, int n_rows, int d,
                                   const int* __restrict__ leaf_start, int n_leaves,
                                   const int* __restrict__ combine_steps) {
  extern __shared__ float leaf_sum[];
  const long r = blockIdx.x;
  if (r >= n_rows) return;
  const int2 de = desc[r];
  exact_generic_row(PqRow{codes + (size_t)de.x * M, codebook, Ks, Ds}, queries + (size_t)de.y * d, d, leaf_start,
                    n_leaves, combine_steps, leaf_sum, rs + r);
}

// d = 256*NCH with 8 leaves (768, 1024), M = 8*MW code bytes per row, Ds % 4 == 0: the float
// exact_lds kernel with the row loads replaced by codeword loads.  Lane t of a wave holds the
// codes of work row t (one 8*MW-byte load per tile); for a row, the code words are broadcast with
// v_readlane and each lane picks the byte of the subspace its 4 elements fall in.
template <int NCH, int MW, int U, int TILE>
__global__ void __launch_bounds__(256)
gather_dot_pq_lds_kernel(const uint8_t* __restrict__ codes, const float* __restrict__ codebook, int Ks, int Ds,
                         const float* __restrict__ queries, const int2* __restrict__ desc,
                         float* __restrict__ rs, int n_rows) {
  constexpr int D = 256 * NCH;
  constexpr int LL = D / 8;
  constexpr int NPER = LL / 8;
  constexpr int LPAD = LL + 8;
  constexpr int M = 8 * MW;
  static_assert(TILE % U == 0 && TILE <= 64, "bad tile");
  __shared__ __attribute__((aligned(16))) float tile[4][8 * LPAD];

  const int lane = threadIdx.x & 63;
  const int wave_in_block = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = blockIdx.x * (blockDim.x >> 6) + wave_in_block;
  const long base = (long)wave * TILE;
  if (base >= n_rows) return;
  const int n_here = (n_rows - base) < TILE ? (int)(n_rows - base) : TILE;
  const int my_t = lane < n_here ? lane : n_here - 1;
  const int2 my = desc[base + my_t];
  uint32_t mycode[2 * MW];  // the M code bytes of work row `lane`
  {
    const uint2* cp = reinterpret_cast<const uint2*>(codes + (size_t)my.x * M);
#pragma unroll
    for (int w = 0; w < MW; ++w) {
      const uint2 t = cp[w];
      mycode[2 * w] = t.x;
      mycode[2 * w + 1] = t.y;
    }
  }

  float* buf = tile[wave_in_block];
  int woff[NCH], wsel[NCH], shift[NCH];
  const float* cbase[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int e = 4 * lane + 256 * c;
    woff[c] = (e / LL) * LPAD + (e % LL);
    const int m = e / Ds;  // the 4 elements e..e+3 lie in one subspace because Ds % 4 == 0
    wsel[c] = m >> 2;
    shift[c] = 8 * (m & 3);
    cbase[c] = codebook + (size_t)m * Ks * Ds + (e - m * Ds);
  }
  const float* rbuf = buf + (lane >> 3) * LPAD + (lane & 7);

  float4 q[NCH];
  int cur_q = -1;
#pragma unroll 1
  for (int b = 0; b < TILE; b += U) {
    if (b >= n_here) break;
    float4 v[U][NCH];
    int qn[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = (b + u) < n_here ? (b + u) : n_here - 1;
      qn[u] = __builtin_amdgcn_readlane(my.y, t);
      uint32_t word[2 * MW];
#pragma unroll
      for (int w = 0; w < 2 * MW; ++w) word[w] = __builtin_amdgcn_readlane(mycode[w], t);
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        uint32_t sel = word[0];
#pragma unroll
        for (int w = 1; w < 2 * MW; ++w) sel = (wsel[c] == w) ? word[w] : sel;
        int code = (sel >> shift[c]) & 0xFF;
        code = code < Ks ? code : Ks - 1;  // a corrupt code must not read past the codebook (uploads reject them; belt and braces)
        v[u][c] = *reinterpret_cast<const float4*>(cbase[c] + (size_t)code * Ds);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (qn[u] != cur_q) {
        cur_q = qn[u];
        const float4* qp = reinterpret_cast<const float4*>(queries + (size_t)cur_q * D) + lane;
#pragma unroll
        for (int c = 0; c < NCH; ++c) q[c] = qp[64 * c];
      }
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        float4 p;
        p.x = __fmul_rn(q[c].x, v[u][c].x);
        p.y = __fmul_rn(q[c].y, v[u][c].y);
        p.z = __fmul_rn(q[c].z, v[u][c].z);
        p.w = __fmul_rn(q[c].w, v[u][c].w);
        *reinterpret_cast<float4*>(buf + woff[c]) = p;
      }
      __builtin_amdgcn_wave_barrier();
      float acc = rbuf[0];
#pragma unroll
      for (int i = 1; i < NPER; ++i) acc = __fadd_rn(acc, rbuf[8 * i]);
      __builtin_amdgcn_wave_barrier();
      acc = butterfly<8>(acc);
      if (lane == 0 && (b + u) < n_here) rs[base + b + u] = acc;
    }
  }
}

// ---- asymmetric distance computation (the "fast" order for PQ rows) --------------------------------
// <q, decode(codes)> = sum_m <q_m, codeword[m][code_m]>: the M partial products per query are
// tabulated once (lut[q][m][ks], Q*M*Ks floats), after which a row costs M table look-ups.
// Different summation order than decode-then-dot, same value to ~1e-6 relative.
// codebook_t is the codebook transposed to [M][Ds][Ks] so that the Ks threads of a block read
// consecutive floats.
__global__ void __launch_bounds__(256)
pq_lut_kernel(const float* __restrict__ queries, const float* __restrict__ codebook_t, int M, int Ks, int Ds,
              float* __restrict__ lut) {
  const int q = blockIdx.x, m = blockIdx.y;
  const float* qv = queries + ((size_t)q * M + m) * Ds;      // wave-uniform: scalar loads
  const float* ct = codebook_t + (size_t)m * Ds * Ks;
  for (int ks = threadIdx.x; ks < Ks; ks += blockDim.x) {
    float acc = 0.f;
    for (int i = 0; i < Ds; ++i) acc = fmaf(qv[i], ct[(size_t)i * Ks + ks], acc);
    lut[((size_t)q * M + m) * Ks + ks] = acc;
  }
}

// ---- exact ADC: when a subspace IS a leaf of NumPy's summation tree --------------------------------
// For d = M*Ds with Ds <= 128 and M a power of two whose subspaces coincide with the leaves of the
// pairwise recursion (d = 768, M = 8: leaves of 96 -- the reference's own NanoPQ(8, 256) on 768-d
// vectors), the sum NumPy forms over one leaf involves one subspace only:
//     leaf_m = tree8( sum_i q[m*Ds + 8i + j] * codeword[m][code_m][8i + j] )
// so it can be tabulated per (query, subspace, codeword) WITH NumPy's rounding, and a row's score
// is the balanced pairwise combination of its M table entries -- bit-identical to
// np.sum(q * decode(codes)) at the cost of M look-ups and M-1 adds per row.
__global__ void __launch_bounds__(256)
pq_lut_exact_kernel(const float* __restrict__ queries, const float* __restrict__ codebook_t, int M, int Ks, int Ds,
                    float* __restrict__ lut) {
  const int q = blockIdx.x, m = blockIdx.y;
  const float* qv = queries + ((size_t)q * M + m) * Ds;
  const float* ct = codebook_t + (size_t)m * Ds * Ks;
  for (int ks = threadIdx.x; ks < Ks; ks += blockDim.x) {
    float res;
    if (Ds < 8) {
      res = -0.0f;
      for (int i = 0; i < Ds; ++i) res = __fadd_rn(res, __fmul_rn(qv[i], ct[(size_t)i * Ks + ks]));
    } else {
      float r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = __fmul_rn(qv[j], ct[(size_t)j * Ks + ks]);
      const int lim = Ds - (Ds % 8);
      for (int i = 8; i < lim; i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], __fmul_rn(qv[i + j], ct[(size_t)(i + j) * Ks + ks]));
      }
      res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                      __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
      for (int i = lim; i < Ds; ++i) res = __fadd_rn(res, __fmul_rn(qv[i], ct[(size_t)i * Ks + ks]));
    }
    lut[((size_t)q * M + m) * Ks + ks] = res;
  }
}

template <int M>
__global__ void __launch_bounds__(256)
pq_adc_exact_kernel(const uint8_t* __restrict__ codes, int Ks, const float* __restrict__ lut,
                    const int2* __restrict__ desc, float* __restrict__ rs, int n_rows) {
  extern __shared__ float table[];
  const long r0 = (long)blockIdx.x * blockDim.x;
  const long r = r0 + threadIdx.x;
  const long last = (r0 + blockDim.x - 1) < (long)n_rows ? (r0 + blockDim.x - 1) : (long)n_rows - 1;
  const int q_first = desc[r0].y, q_last = desc[last].y;
  const bool one_query = q_first == q_last;
  if (one_query) {
    const float* src = lut + (size_t)q_first * M * Ks;
    for (int i = threadIdx.x; i < M * Ks; i += blockDim.x) table[i] = src[i];
  }
  __syncthreads();
  if (r >= n_rows) return;
  const int2 de = desc[r];
  const uint8_t* c = codes + (size_t)de.x * M;
  const float* t = one_query ? table : lut + (size_t)de.y * M * Ks;
  float leaf[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int code = c[m] < Ks ? c[m] : Ks - 1;  // never past the table (uploads reject such codes)
    leaf[m] = t[(size_t)m * Ks + code];
  }
  // balanced pairwise combination = NumPy's recursion over equal leaves
#pragma unroll
  for (int s = 1; s < M; s *= 2) {
#pragma unroll
    for (int i = 0; i + s < M; i += 2 * s) leaf[i] = __fadd_rn(leaf[i], leaf[i + s]);
  }
  rs[r] = leaf[0];
}

// Exact ADC straight from the candidate lists (single-row candidates: PASSAGE, FIRSTP, or one passage per
// document): candidate -> row -> codes -> table look-ups -> dense score -> interpolation in ONE kernel.  A
// PQ row is 8 code bytes; the {row, query} descriptor the general pipeline writes and reads back per row is
// 8 bytes too, and the interpolation pass re-reads the scores -- fused, none of that traffic exists.
// q_of(p): the query of pair p (proportional guess, binary-search fallback: aux_kernels.hip.h has the same).
template <int M>
__global__ void __launch_bounds__(256)
pq_adc_exact_pairs_kernel(const uint8_t* __restrict__ codes, int Ks, const float* __restrict__ lut,
                          const int32_t* __restrict__ pair_ids, const int64_t* __restrict__ pair_off, int n_queries,
                          double queries_per_pair, const int32_t* __restrict__ row_of, const float* __restrict__ sparse,
                          float a, float b, float* __restrict__ ff, float* __restrict__ interp, long n_pairs, long n_index_rows) {
  extern __shared__ float table[];
  auto q_of = [&](long p) -> int {
    int q = (int)((double)p * queries_per_pair);
    q = q < 0 ? 0 : (q > n_queries - 1 ? n_queries - 1 : q);
    if (pair_off[q] <= p && p < pair_off[q + 1]) return q;
    int lo = 0, hi = n_queries;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (pair_off[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
  };
  const long p0 = (long)blockIdx.x * blockDim.x;
  const long p = p0 + threadIdx.x;
  const long last = (p0 + blockDim.x - 1) < n_pairs ? (p0 + blockDim.x - 1) : n_pairs - 1;
  const int q_first = q_of(p0), q_last = q_of(last);  // block-uniform
  const bool one_query = q_first == q_last;
  if (one_query) {
    const float* src = lut + (size_t)q_first * M * Ks;
    for (int i = threadIdx.x; i < M * Ks; i += blockDim.x) table[i] = src[i];
  }
  __syncthreads();
  if (p >= n_pairs) return;
  const int q = one_query ? q_first : q_of(p);
  const int32_t o = pair_ids[p];
  int32_t row = row_of ? row_of[o] : o;
  row = (uint32_t)row < (unsigned long)n_index_rows ? row : 0;  // see expand_single_kernel
  const uint8_t* c = codes + (size_t)row * M;
  const float* t = one_query ? table : lut + (size_t)q * M * Ks;
  float leaf[M];
#pragma unroll
  for (int m = 0; m < M; ++m) {
    const int code = c[m] < Ks ? c[m] : Ks - 1;
    leaf[m] = t[(size_t)m * Ks + code];
  }
#pragma unroll
  for (int s = 1; s < M; s *= 2) {
#pragma unroll
    for (int i = 0; i + s < M; i += 2 * s) leaf[i] = __fadd_rn(leaf[i], leaf[i + s]);
  }
  ff[p] = leaf[0];
  if (sparse != nullptr) interp[p] = __fadd_rn(__fmul_rn(a, sparse[p]), __fmul_rn(b, leaf[0]));
}

// ... and the per-query ordering folded in as well (ranking.py:115-117): ONE workgroup per query stages the query's table,
// scores its candidates (each thread IPT of them: the ordinals first, then the code words, then the look-ups, so that the
// dependent loads of all of them are in flight together), keeps the order-preserving images of the keys in LDS and orders
// them there with the block-level radix sort block_sort_desc_kernel uses -- candidate -> row -> codes -> look-ups -> dense
// score -> interpolation -> order in one launch, nothing written that is read back.  Stable, descending, ties by position,
// -0.0 counted as +0.0: bit for bit the order of the separate sort (test_pq_* and test_per_query_order).
template <int M, unsigned BS, unsigned IPT>
__global__ void __launch_bounds__(BS)
pq_adc_exact_pairs_sort_kernel(const uint8_t* __restrict__ codes, int Ks, const float* __restrict__ lut, const int32_t* __restrict__ pair_ids,
                               const int64_t* __restrict__ pair_off, const int32_t* __restrict__ row_of, const float* __restrict__ sparse, float a, float b,
                               int sort_ff, float* __restrict__ ff, float* __restrict__ interp, int32_t base, int32_t* __restrict__ order, long n_index_rows) {
  using Sort = rocprim::block_radix_sort<uint32_t, BS, IPT, int32_t>;
  __shared__ typename Sort::storage_type storage;
  __shared__ uint32_t stage[BS * IPT];
  extern __shared__ float table[];
  const int q = blockIdx.x;
  const int64_t lo = pair_off[q];
  const int n = (int)(pair_off[q + 1] - lo);
  if (n == 0) return;  // block-uniform
  {
    const float* __restrict__ src = lut + (size_t)q * M * Ks;
    for (int i = threadIdx.x; i < M * Ks; i += BS) table[i] = src[i];
  }
  __syncthreads();
  int32_t row[IPT];
#pragma unroll
  for (unsigned j = 0; j < IPT; ++j) {
    const int i = (int)(threadIdx.x + j * BS);
    int32_t r = 0;
    if (i < n) {
      const int32_t o = pair_ids[lo + i];
      r = row_of ? row_of[o] : o;
      r = (uint32_t)r < (unsigned long)n_index_rows ? r : 0;  // see expand_single_kernel
    }
    row[j] = r;
  }
  static_assert(M % 8 == 0, "the code words are fetched eight at a time");
  uint2 word[IPT][M / 8];
#pragma unroll
  for (unsigned j = 0; j < IPT; ++j) {
    const uint2* __restrict__ c = reinterpret_cast<const uint2*>(codes + (size_t)row[j] * M);
#pragma unroll
    for (int w = 0; w < M / 8; ++w) word[j][w] = c[w];
  }
#pragma unroll
  for (unsigned j = 0; j < IPT; ++j) {
    const int i = (int)(threadIdx.x + j * BS);
    uint32_t image = 0u;
    if (i < n) {
      float leaf[M];
#pragma unroll
      for (int m = 0; m < M; ++m) {
        const uint32_t half = (m & 4) ? word[j][m / 8].y : word[j][m / 8].x;
        int code = (int)((half >> (8 * (m & 3))) & 0xFFu);
        code = code < Ks ? code : Ks - 1;
        leaf[m] = table[m * Ks + code];
      }
#pragma unroll
      for (int s = 1; s < M; s *= 2) {
#pragma unroll
        for (int t = 0; t + s < M; t += 2 * s) leaf[t] = __fadd_rn(leaf[t], leaf[t + s]);
      }
      float key = leaf[0];
      ff[lo + i] = key;
      if (sparse != nullptr) {
        const float mixed = __fadd_rn(__fmul_rn(a, sparse[lo + i]), __fmul_rn(b, key));
        interp[lo + i] = mixed;
        if (!sort_ff) key = mixed;
      }
      uint32_t bits = __float_as_uint(key);
      if (bits == 0x80000000u) bits = 0u;
      image = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
    }
    stage[i] = image;
  }
  __syncthreads();
  uint32_t k[IPT];
  int32_t v[IPT];
#pragma unroll
  for (unsigned j = 0; j < IPT; ++j) {
    const int at = (int)(threadIdx.x * IPT + j);
    k[j] = stage[at];
    v[j] = at;
  }
  __syncthreads();
  Sort().sort_desc(k, v, storage);
  __syncthreads();
#pragma unroll
  for (unsigned j = 0; j < IPT; ++j) stage[threadIdx.x * IPT + j] = (uint32_t)v[j];
  __syncthreads();
  const int32_t first = base + (int32_t)lo;
  for (int i = threadIdx.x; i < n; i += BS) order[lo + i] = first + (int32_t)stage[i];
}

// One thread per row.  Rows of a query are contiguous, so almost every block works for a single
// query: its M*Ks-float table is staged in LDS once and the M look-ups per row hit LDS.
__global__ void __launch_bounds__(256)
pq_adc_kernel(const uint8_t* __restrict__ codes, int M, int Ks, const float* __restrict__ lut,
              const int2* __restrict__ desc, float* __restrict__ rs, int n_rows, int stage) {
  extern __shared__ float table[];
  const long r0 = (long)blockIdx.x * blockDim.x;
  const long r = r0 + threadIdx.x;
  const long last = (r0 + blockDim.x - 1) < (long)n_rows ? (r0 + blockDim.x - 1) : (long)n_rows - 1;
  const int q_first = desc[r0].y, q_last = desc[last].y;   // block-uniform
  // stage == 0: the M*Ks table does not fit the LDS the launch was given; read it through L2
  const bool one_query = stage && q_first == q_last;
  if (one_query) {
    const float* src = lut + (size_t)q_first * M * Ks;
    for (int i = threadIdx.x; i < M * Ks; i += blockDim.x) table[i] = src[i];
  }
  __syncthreads();
  if (r >= n_rows) return;
  const int2 de = desc[r];
  const uint8_t* c = codes + (size_t)de.x * M;
  float acc = 0.f;
  const int top = Ks - 1;  // never past the table (uploads reject codes >= Ks)
  if (one_query) {
    for (int m = 0; m < M; ++m) acc += table[m * Ks + (c[m] < Ks ? c[m] : top)];
  } else {
    const float* t = lut + (size_t)de.y * M * Ks;
    for (int m = 0; m < M; ++m) acc += t[(size_t)m * Ks + (c[m] < Ks ? c[m] : top)];
  }
  rs[r] = acc;
}

// first position whose code byte is >= Ks (atomicMin into *first_bad, preset to ~0): run after
// every upload into a product-quantized index with Ks < 256
__global__ void __launch_bounds__(256)
pq_check_codes_kernel(const uint8_t* __restrict__ codes, long n_bytes, int Ks, unsigned long long* __restrict__ first_bad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_bytes && codes[i] >= Ks) atomicMin(first_bad, (unsigned long long)i);
}

}  // namespace ffx
