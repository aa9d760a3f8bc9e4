// host_io.cpp -- host-side natives of the re-ranking path: string-id tables and TREC run files.
//
// Once the arithmetic runs at HBM speed the reference's host work dominates end to end: the
// per-id Python loop of get_indices (src/fast_forward/index/util.py:29-41), the pandas string
// merges (index/base.py:296-298, :422) and read_csv / to_csv of the run
// (src/fast_forward/ranking.py:348-366, :388-409).  This file replaces those with
//   ffx_ids_*  an insertion-ordered string -> ordinal table (open addressing, FNV-1a)
//   ffx_run_*  a TREC run parsed straight into the CSR the kernels take, and written back
// No device code here; plain C++17.
#include <atomic>
#include <cctype>
#include <cmath>
#include <numeric>
#include <system_error>

#include "host_util.h"
#include "string_code.h"

using namespace ffx_host;

namespace ffx_host {
std::string& stage_report() {
  static thread_local std::string report;
  return report;
}
}  // namespace ffx_host

namespace {

// bits[r / 64] |= 1 << (r % 64) for every row number in rows[0 .. n) (all valid): threads take blocks of the list, the
// updates are atomic.  (Tried: no atomics, every thread reading the whole list and setting the bits of its own share of the
// words -- on the 16 cores of the GPU box 26 ms against 15 ms for 4.5 M rows; the atomic form wins there.)
template <class Int>
void mark_rows(uint64_t* bits, const Int* rows, int64_t n, int T) {
  run_threads(T, [&](int t) {
    for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i)
      __atomic_fetch_or(&bits[(int64_t)rows[i] >> 6], 1ull << ((int64_t)rows[i] & 63), __ATOMIC_RELAXED);
  });
}

inline std::string_view fixed_width(const char* base, int32_t width, int64_t i) {
  const char* p = base + (size_t)i * width;
  size_t n = 0;
  while (n < (size_t)width && p[n] != '\0') ++n;
  return {p, n};
}

}  // namespace

extern "C" {

int ffx_ids_create(ffx_ids** out) {
  if (!out) return ffx_fail(FFX_BAD_ARG, "out is NULL");
  *out = new (std::nothrow) ffx_ids();
  return *out ? FFX_OK : ffx_fail(FFX_OOM, "host allocation failed");
}

int ffx_ids_destroy(ffx_ids* t) {
  delete t;
  return FFX_OK;
}

int ffx_ids_count(const ffx_ids* t, int64_t* n) {
  if (!t || !n) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  *n = t->size();
  return FFX_OK;
}

int ffx_ids_intern(ffx_ids* t, const char* strs, int32_t width, int64_t n, int32_t* out_ordinals) {
  if (!t || (n > 0 && !strs) || width <= 0 || n < 0) return ffx_fail(FFX_BAD_ARG, "bad argument");
  try {
    for (int64_t i = 0; i < n; ++i) {
      if (t->size() >= INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "too many distinct ids for int32 ordinals");
      const int32_t o = t->intern(fixed_width(strs, width, i));
      if (out_ordinals) out_ordinals[i] = o;
    }
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  }
  return FFX_OK;
}

int ffx_ids_lookup(const ffx_ids* t, const char* strs, int32_t width, int64_t n, int32_t* out_ordinals,
                   int64_t* first_missing) {
  if (!t || (n > 0 && (!strs || !out_ordinals)) || width <= 0 || n < 0) return ffx_fail(FFX_BAD_ARG, "bad argument");
  int64_t miss = -1;
  for (int64_t i = 0; i < n; ++i) {
    const int32_t o = t->find(fixed_width(strs, width, i));
    out_ordinals[i] = o;
    if (o < 0 && miss < 0) miss = i;
  }
  if (first_missing) *first_missing = miss;
  return miss < 0 ? FFX_OK : FFX_NOT_FOUND;
}

// The same look-up for strings handed over as ONE buffer with a separator byte between them
// (what "\n".join(ids).encode() builds far faster than a fixed-width copy of every string).
// Exactly n fields must be present.  Large requests are split over a few threads (the table is
// only read).
int ffx_ids_lookup_packed(const ffx_ids* t, const char* data, int64_t n_bytes, char sep, int64_t n,
                          int32_t* out_ordinals, int64_t* first_missing) {
  if (first_missing) *first_missing = -1;
  if (!t || n < 0 || n_bytes < 0 || (n > 0 && (!data || !out_ordinals))) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n == 0) return n_bytes == 0 ? FFX_OK : ffx_fail(FFX_BAD_ARG, "bytes given for zero strings");
  // Threads take byte ranges of the buffer: the separators of every range are counted first (that numbers the strings),
  // then every thread looks up the strings that START in its range, eight at a time -- a look-up is three dependent
  // cache misses (slot, offsets, bytes of the candidate) in a table of millions, and eight of each are in flight at once.
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::max(1, std::min(32, usable_cores())), n >> 15, n_bytes >> 16}));
  std::vector<int64_t> seps((size_t)T + 1, 0);
  auto range_lo = [&](int k) { return n_bytes * k / T; };
  auto count = [&](int k) {
    int64_t c = 0;
    const char *p = data + range_lo(k), *end = data + range_lo(k + 1);
    while (p < end) {
      const char* q = static_cast<const char*>(memchr(p, sep, (size_t)(end - p)));
      if (!q) break;
      ++c;
      p = q + 1;
    }
    seps[(size_t)k + 1] = c;
  };
  std::atomic<int64_t> miss{INT64_MAX};
  auto lookup = [&](int k) {
    // a string belongs to the range that holds the separator in front of it (string 0 to range 0): range k owns strings
    // first .. first + owned - 1
    const int64_t first = seps[(size_t)k] + (k == 0 ? 0 : 1);
    const int64_t owned = seps[(size_t)k + 1] - seps[(size_t)k] + (k == 0 ? 1 : 0);
    if (owned <= 0) return;
    const char* const buf_end = data + n_bytes;
    const char* p = data + range_lo(k);
    if (k > 0) p = static_cast<const char*>(memchr(p, sep, (size_t)(buf_end - p))) + 1;  // (there is one: owned > 0)
    int64_t mine = INT64_MAX;
    constexpr int B = 8;
    std::string_view sv[B];
    uint64_t h[B];
    for (int64_t done = 0; done < owned;) {
      const int nb = (int)std::min<int64_t>(B, owned - done);
      for (int b2 = 0; b2 < nb; ++b2) {
        const char* q = static_cast<const char*>(memchr(p, sep, (size_t)(buf_end - p)));
        const char* e = q ? q : buf_end;
        sv[b2] = std::string_view(p, (size_t)(e - p));
        h[b2] = fnv1a(p, (size_t)(e - p));
        __builtin_prefetch(&t->slots[h[b2] & t->mask]);
        p = e + 1;
      }
      for (int b2 = 0; b2 < nb; ++b2) {  // (a slot says whose string it holds: the right one, almost always)
        const uint64_t v = t->slots[h[b2] & t->mask];
        if (v != 0) __builtin_prefetch(&t->off[(size_t)((uint32_t)v - 1)]);
      }
      for (int b2 = 0; b2 < nb; ++b2) {
        const uint64_t v = t->slots[h[b2] & t->mask];
        if (v != 0) __builtin_prefetch(t->arena.data() + t->off[(size_t)((uint32_t)v - 1)]);
      }
      for (int b2 = 0; b2 < nb; ++b2) {
        const int32_t o = t->find_hashed(sv[b2], h[b2]);
        out_ordinals[first + done + b2] = o;
        if (o < 0 && mine == INT64_MAX) mine = first + done + b2;
      }
      done += nb;
    }
    int64_t seen = miss.load();
    while (mine < seen && !miss.compare_exchange_weak(seen, mine)) {
    }
  };
  try {
    run_threads(T, count);
    for (int k = 0; k < T; ++k) seps[(size_t)k + 1] += seps[(size_t)k];
    if (seps[(size_t)T] != n - 1) return ffx_fail(FFX_BAD_ARG, "the buffer does not hold exactly %lld separated strings", (long long)n);
    run_threads(T, lookup);
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
  const int64_t m = miss.load();
  if (m == INT64_MAX) return FFX_OK;
  if (first_missing) *first_missing = m;
  return FFX_NOT_FOUND;
}

// ffx_ids_intern for n strings handed over as one buffer with the byte `sep` between them (what
// "\n".join(ids).encode() builds in a fraction of the time a fixed-width copy takes): Index.add at
// millions of ids per call (memory.py:85-99 does a dict insert per id).
int ffx_ids_intern_packed(ffx_ids* t, const char* data, int64_t n_bytes, char sep, int64_t n, int32_t* out_ordinals) {
  if (!t || n < 0 || n_bytes < 0 || (n > 0 && !data)) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n == 0) return n_bytes == 0 ? FFX_OK : ffx_fail(FFX_BAD_ARG, "bytes given for zero strings");
  // validate the field count before touching the table: nothing is modified on a malformed buffer
  int64_t seps = 0;
  for (const char* p = data; (p = static_cast<const char*>(memchr(p, sep, (size_t)(data + n_bytes - p)))) != nullptr; ++p) ++seps;
  if (seps != n - 1) return ffx_fail(FFX_BAD_ARG, "the buffer does not hold exactly %lld separated strings", (long long)n);
  if (t->size() + n >= INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "too many distinct ids for int32 ordinals");
  try {
    t->expect(n, n_bytes);
    const char* p = data;
    const char* end = data + n_bytes;
    for (int64_t i = 0; i < n; ++i) {
      const char* q = static_cast<const char*>(memchr(p, sep, (size_t)(end - p)));
      if (!q) q = end;
      const int32_t o = t->intern(std::string_view(p, (size_t)(q - p)));
      if (out_ordinals) out_ordinals[i] = o;
      p = q + 1;
    }
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  }
  return FFX_OK;
}

// Numbering the strings of one joined buffer WITHOUT keeping a table (Ranking.from_file on a large run: the
// factorisation of the id column, ranking.py:95-98 / pandas.factorize): codes[i] = number of string i, equal
// strings get equal numbers, numbers are dense 0..n_names-1; *names_blob receives the distinct strings in
// number order joined by `sep` (released with ffx_buffer_free).  Parallel where a table insert per string is
// serial: offsets and hashes by row ranges, then the rows are dealt into 256 shards by hash and every shard is
// numbered on its own (open addressing over the shard's rows, then the shard's distinct strings put in (hash, bytes)
// order), so the numbering depends on the SET of strings only: not on the row order, not on the thread count.
int ffx_strings_code(const char* data, int64_t n_bytes, char sep, int64_t n, int32_t n_threads, int32_t* codes, char** names_blob,
                     int64_t* names_bytes, int64_t* n_names) {
  if (names_blob) *names_blob = nullptr;
  if (names_bytes) *names_bytes = 0;
  if (n_names) *n_names = 0;
  if (n < 0 || n_bytes < 0 || (n > 0 && (!data || !codes)) || !names_blob || !names_bytes || !n_names) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n == 0) return n_bytes == 0 ? FFX_OK : ffx_fail(FFX_BAD_ARG, "bytes given for zero strings");
  if (n >= INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "too many strings for int32 codes");
  int T = n_threads > 0 ? n_threads : usable_cores();
  T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)64, n / (1 << 16) + 1}));
  auto in_threads = [&](auto&& body) { run_threads(T, body); };  // body(t) for t in 0..T-1
  try {
    // 1. where every string starts: separators counted per byte range, then written per byte range
    std::vector<int64_t> seps_in((size_t)T + 1, 0);
    auto byte_range = [&](int t) { return std::pair<int64_t, int64_t>(n_bytes * t / T, n_bytes * (t + 1) / T); };
    in_threads([&](int t) {
      const auto [lo, hi] = byte_range(t);
      int64_t c = 0;
      for (const char* p = data + lo; (p = static_cast<const char*>(memchr(p, sep, (size_t)(data + hi - p)))) != nullptr; ++p) ++c;
      seps_in[(size_t)t + 1] = c;
    });
    for (int t = 0; t < T; ++t) seps_in[(size_t)t + 1] += seps_in[(size_t)t];
    if (seps_in[(size_t)T] != n - 1) return ffx_fail(FFX_BAD_ARG, "the buffer does not hold exactly %lld separated strings", (long long)n);
    std::vector<int64_t> start((size_t)n + 1);
    start[0] = 0;
    start[(size_t)n] = n_bytes + 1;  // string i is data[start[i] .. start[i+1] - 1)
    in_threads([&](int t) {
      const auto [lo, hi] = byte_range(t);
      int64_t i = seps_in[(size_t)t] + 1;
      for (const char* p = data + lo; (p = static_cast<const char*>(memchr(p, sep, (size_t)(data + hi - p)))) != nullptr; ++p) start[(size_t)i++] = (p - data) + 1;
    });
    auto view = [&](int64_t i) { return std::string_view(data + start[(size_t)i], (size_t)(start[(size_t)i + 1] - 1 - start[(size_t)i])); };
    return strings_code_views(n, T, view, sep, codes, names_blob, names_bytes, n_names);
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
}

// ffx_strings_code for strings that lie back to back in `data` with their boundaries given: string i is
// data[offsets[i] .. offsets[i+1]) (what an Arrow string array holds; no separator inside `data`, none needed).
// `sep` only joins the distinct strings in *names_blob.
int ffx_strings_code_offsets(const char* data, const int64_t* offsets, int64_t n, char sep, int32_t n_threads, int32_t* codes, char** names_blob,
                             int64_t* names_bytes, int64_t* n_names) {
  if (names_blob) *names_blob = nullptr;
  if (names_bytes) *names_bytes = 0;
  if (n_names) *n_names = 0;
  if (n < 0 || (n > 0 && (!offsets || !codes)) || !names_blob || !names_bytes || !n_names) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n == 0) return FFX_OK;
  if (n >= INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "too many strings for int32 codes");
  if (offsets[0] < 0 || (offsets[n] > offsets[0] && !data)) return ffx_fail(FFX_BAD_ARG, "bad offsets");
  int T = n_threads > 0 ? n_threads : usable_cores();
  T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)64, n / (1 << 16) + 1}));
  try {
    std::atomic<bool> bad(false);
    run_threads(T, [&](int t) {
      for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i)
        if (offsets[i + 1] < offsets[i]) bad.store(true);
    });
    if (bad.load()) return ffx_fail(FFX_BAD_ARG, "offsets are not monotone");
    auto view = [&](int64_t i) { return std::string_view(data + offsets[i], (size_t)(offsets[i + 1] - offsets[i])); };
    StageClock clock;
    return strings_code_views(n, T, view, sep, codes, names_blob, names_bytes, n_names, &clock);
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
}

// ffx_strings_code_offsets for 32-bit offsets (an Arrow `string` array; `large_string` has 64-bit ones)
int ffx_strings_code_offsets32(const char* data, const int32_t* offsets, int64_t n, char sep, int32_t n_threads, int32_t* codes, char** names_blob,
                               int64_t* names_bytes, int64_t* n_names) {
  if (names_blob) *names_blob = nullptr;
  if (names_bytes) *names_bytes = 0;
  if (n_names) *n_names = 0;
  if (n < 0 || (n > 0 && (!offsets || !codes)) || !names_blob || !names_bytes || !n_names) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n == 0) return FFX_OK;
  if (n >= INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "too many strings for int32 codes");
  if (offsets[0] < 0 || (offsets[n] > offsets[0] && !data)) return ffx_fail(FFX_BAD_ARG, "bad offsets");
  const int T = threads_for(n, 1 << 16, n_threads);
  try {
    std::atomic<bool> bad(false);
    run_threads(T, [&](int t) {
      for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i)
        if (offsets[i + 1] < offsets[i]) bad.store(true);
    });
    if (bad.load()) return ffx_fail(FFX_BAD_ARG, "offsets are not monotone");
    auto view = [&](int64_t i) { return std::string_view(data + offsets[i], (size_t)(offsets[i + 1] - offsets[i])); };
    StageClock clock;
    return strings_code_views(n, T, view, sep, codes, names_blob, names_bytes, n_names, &clock);
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
}

// ffx_strings_code for n NUL-padded fields of `width` bytes each (numpy 'S<width>': the id columns of an index file,
// a column of fixed-width ids); a field ends at its first NUL byte
int ffx_strings_code_fixed(const char* data, int32_t width, int64_t n, char sep, int32_t n_threads, int32_t* codes, char** names_blob,
                           int64_t* names_bytes, int64_t* n_names) {
  if (names_blob) *names_blob = nullptr;
  if (names_bytes) *names_bytes = 0;
  if (n_names) *n_names = 0;
  if (n < 0 || width <= 0 || (n > 0 && (!data || !codes)) || !names_blob || !names_bytes || !n_names) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n == 0) return FFX_OK;
  if (n >= INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "too many strings for int32 codes");
  const int T = threads_for(n, 1 << 16, n_threads);
  try {
    auto view = [&](int64_t i) { return fixed_width(data, width, i); };
    return strings_code_views(n, T, view, sep, codes, names_blob, names_bytes, n_names);
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
}

// The first group of rows (group_off[g] .. group_off[g+1]) that holds one code twice, -1 in *group if there is none: the
// repeated (query, id) pair of ranking.py:95-98 for rows that stand grouped by query and hold id codes in [0, n_codes).
int ffx_host_repeated_in_groups(const int32_t* codes, const int64_t* group_off, int64_t n_groups, int64_t n_codes, int32_t n_threads, int64_t* group) {
  if (!group) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  *group = -1;
  if (n_groups < 0 || n_codes < 0 || (n_groups > 0 && !group_off)) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n_groups == 0) return FFX_OK;
  if (group_off[0] != 0) return ffx_fail(FFX_BAD_ARG, "group_off[0] must be 0");
  for (int64_t g = 0; g < n_groups; ++g)
    if (group_off[g + 1] < group_off[g]) return ffx_fail(FFX_BAD_ARG, "group_off not monotone at %lld", (long long)g);
  const int64_t n = group_off[n_groups];
  if (n > 0 && !codes) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  const int T = threads_for(n, 1 << 18, n_threads);
  try {
    std::atomic<int64_t> bad(INT64_MAX);
    run_threads(T, [&](int t) {
      for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i)
        if ((uint32_t)codes[i] >= (uint64_t)n_codes) {
          int64_t cur = bad.load();
          while (i < cur && !bad.compare_exchange_weak(cur, i)) {
          }
          return;
        }
    });
    if (bad.load() != INT64_MAX) return ffx_fail(FFX_BAD_ARG, "row %lld holds code %d, outside the %lld codes", (long long)bad.load(), codes[bad.load()], (long long)n_codes);
    *group = first_group_with_repeated_code(codes, group_off, n_groups, n_codes, T);
    return FFX_OK;
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
}

// Numbering n 8-byte values (equal values get equal numbers, numbers are dense): the same sharded scheme as
// ffx_strings_code with the value itself as the key.  What it is for: a column of a pandas frame is an array of
// POINTERS to Python strings, and a run of millions of rows names each query / document through the same few
// objects (pandas' readers intern the strings of a column; frames built from arrays of names share them by
// construction) -- numbering the pointers first leaves only the DISTINCT objects to be read as strings
// (Ranking.__init__, ranking.py:95-116, on 34.9 M rows).  Numbers ascend with the values (they depend on the set of
// values only); first_row[c] = a row holding number c; first_row must hold n entries at most.
int ffx_host_code_u64(const uint64_t* values, int64_t n, int32_t n_threads, int32_t* codes, int64_t* first_row, int64_t* n_distinct) {
  if (n_distinct) *n_distinct = 0;
  if (n < 0 || !n_distinct || (n > 0 && (!values || !codes || !first_row))) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n == 0) return FFX_OK;
  if (n >= INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "too many values for int32 codes");
  int T = n_threads > 0 ? n_threads : usable_cores();
  T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)64, n / (1 << 16) + 1}));
  // A column of long RUNS -- the query-id and query-text columns of a run name each query through one object, k rows in a row --
  // is numbered through its run heads: one streaming pass finds them, the few heads are numbered by the general scheme below
  // (the numbers depend on the set of values only, so they are the numbers the whole column would get), and every row takes its
  // head's number.  (Two of the three text columns of a 34.9 M-row run: ~0.2 s each by the general scheme.)
  if (n >= (1 << 16)) {
    try {
      std::vector<std::vector<int64_t>> heads_of((size_t)T);
      run_threads(T, [&](int t) {
        auto& mine = heads_of[(size_t)t];
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        for (int64_t i = lo; i < hi; ++i)
          if (i == 0 || values[i] != values[i - 1]) {
            if ((int64_t)mine.size() > (hi - lo) / 64 + 16) return;  // no long runs here: give up early
            mine.push_back(i);
          }
      });
      int64_t H = 0;
      bool long_runs = true;
      for (int t = 0; t < T; ++t) {
        const int64_t span = n * (t + 1) / T - n * t / T;
        long_runs = long_runs && (int64_t)heads_of[(size_t)t].size() <= span / 64 + 16;
        H += (int64_t)heads_of[(size_t)t].size();
      }
      if (long_runs && H <= n / 64) {
        std::vector<int64_t> head((size_t)H + 1);
        {
          int64_t w = 0;
          for (int t = 0; t < T; ++t)
            for (const int64_t i : heads_of[(size_t)t]) head[(size_t)w++] = i;
          head[(size_t)H] = n;
        }
        std::vector<uint64_t> head_value((size_t)H);
        for (int64_t h = 0; h < H; ++h) head_value[(size_t)h] = values[head[(size_t)h]];
        std::vector<int32_t> head_code((size_t)H);
        std::vector<int64_t> head_first((size_t)H);
        int64_t nd = 0;
        const int rc = ffx_host_code_u64(head_value.data(), H, 1, head_code.data(), head_first.data(), &nd);  // (H < 2^16 or no runs: the general scheme)
        if (rc != FFX_OK) return rc;
        run_threads(T, [&](int t) {
          for (int64_t h = H * t / T; h < H * (t + 1) / T; ++h) std::fill(codes + head[(size_t)h], codes + head[(size_t)h + 1], head_code[(size_t)h]);
        });
        for (int64_t c = 0; c < nd; ++c) first_row[c] = head[(size_t)head_first[(size_t)c]];
        *n_distinct = nd;
        return FFX_OK;
      }
    } catch (const std::bad_alloc&) {
      return ffx_fail(FFX_OOM, "host allocation failed");
    } catch (const std::system_error&) {
      return ffx_fail(FFX_OOM, "could not start host threads");
    }
  }
  constexpr int S = 256;
  auto mix = [](uint64_t v) {
    v ^= v >> 33;
    v *= 0xff51afd7ed558ccdull;
    v ^= v >> 33;
    v *= 0xc4ceb9fe1a85ec53ull;
    return v ^ (v >> 33);
  };
  try {
    // Shards are RANGES of values (splitters from a sample), and inside a shard the distinct values are numbered in
    // ascending order: the numbers come out in ascending order of the values altogether.  For object addresses that is
    // the order in which the objects lie in memory -- whoever reads the distinct objects next (first_row) walks the heap
    // front to back instead of jumping around in it (8.7 M strings: a cache miss per object less, twice) -- and the
    // numbering depends on the set of values only.
    std::vector<uint64_t> split;
    {
      const int64_t m = std::min<int64_t>(n, 1 << 16);
      std::vector<uint64_t> sample((size_t)m);
      for (int64_t i = 0; i < m; ++i) sample[(size_t)i] = values[(size_t)((__int128)i * n / m)];
      std::sort(sample.begin(), sample.end());
      for (int s = 1; s < S; ++s) split.push_back(sample[(size_t)((int64_t)s * m / S)]);
    }
    auto shard_of = [&](uint64_t v) { return (int)(std::upper_bound(split.begin(), split.end(), v) - split.begin()); };
    auto row_range = [&](int t) { return std::pair<int64_t, int64_t>(n * t / T, n * (t + 1) / T); };
    std::vector<uint8_t> shard((size_t)n);
    std::vector<int64_t> count((size_t)T * S, 0);
    run_threads(T, [&](int t) {
      const auto [lo, hi] = row_range(t);
      int64_t* c = count.data() + (size_t)t * S;
      for (int64_t i = lo; i < hi; ++i) ++c[shard[(size_t)i] = (uint8_t)shard_of(values[i])];
    });
    std::vector<int64_t> shard_off((size_t)S + 1, 0);
    {
      int64_t at = 0;
      for (int s = 0; s < S; ++s) {
        shard_off[(size_t)s] = at;
        for (int t = 0; t < T; ++t) {
          const int64_t c = count[(size_t)t * S + s];
          count[(size_t)t * S + s] = at;
          at += c;
        }
      }
      shard_off[(size_t)S] = at;
    }
    // every shard gets its (row, value) pairs side by side: what follows reads them front to back, keeps a shard's table and its
    // distinct values in cache, and touches `codes` -- 4 bytes somewhere in 140 MB per row -- ONCE, when the final numbers are
    // known (the first form looked values[row] and codes[row] up per row in three passes: four cache misses a row, 1.4 of the
    // 2.5 s a frame of 34.9 M Python strings took to number on eight cores)
    std::vector<int32_t> rows((size_t)n);
    std::vector<uint64_t> vals((size_t)n);
    run_threads(T, [&](int t) {
      const auto [lo, hi] = row_range(t);
      int64_t* at = count.data() + (size_t)t * S;
      for (int64_t i = lo; i < hi; ++i) {
        const int64_t w = at[shard[(size_t)i]]++;
        rows[(size_t)w] = (int32_t)i;
        vals[(size_t)w] = values[i];
      }
    });
    std::vector<uint8_t>().swap(shard);
    std::vector<int32_t> code_k((size_t)n);  // the number of the pair at position k, inside its shard
    std::vector<std::vector<int32_t>> firsts((size_t)S);
    std::atomic<int> next_shard(0);
    std::atomic<bool> oom(false);
    run_threads(T, [&](int) {
      try {
        std::vector<int32_t> slot, by_value, renumber;
        std::vector<uint64_t> first_val;
        for (;;) {
          const int s = next_shard.fetch_add(1);
          if (s >= S) return;
          const int64_t lo = shard_off[(size_t)s], hi = shard_off[(size_t)s + 1];
          size_t cap = 16;
          while (cap < (size_t)(hi - lo) * 2) cap <<= 1;
          slot.assign(cap, -1);
          first_val.clear();
          auto& first = firsts[(size_t)s];
          for (int64_t k = lo; k < hi; ++k) {
            const uint64_t v = vals[(size_t)k];
            size_t at = (size_t)(mix(v) >> 8) & (cap - 1);
            for (;;) {
              const int32_t have = slot[at];
              if (have < 0) {
                slot[at] = (int32_t)first.size();
                code_k[(size_t)k] = (int32_t)first.size();
                first.push_back(rows[(size_t)k]);  // (pairs stand in row order inside a shard: the first occurrence)
                first_val.push_back(v);
                break;
              }
              if (first_val[(size_t)have] == v) {
                code_k[(size_t)k] = have;
                break;
              }
              at = (at + 1) & (cap - 1);
            }
          }
          // the shard's distinct values in ascending order
          by_value.resize(first.size());
          renumber.resize(first.size());
          for (size_t j = 0; j < first.size(); ++j) by_value[j] = (int32_t)j;
          std::sort(by_value.begin(), by_value.end(), [&](int32_t x, int32_t y) { return first_val[(size_t)x] < first_val[(size_t)y]; });
          for (size_t j = 0; j < first.size(); ++j) renumber[(size_t)by_value[j]] = (int32_t)j;
          std::vector<int32_t> sorted(first.size());
          for (size_t j = 0; j < first.size(); ++j) sorted[j] = first[(size_t)by_value[j]];
          first.swap(sorted);
          for (int64_t k = lo; k < hi; ++k) code_k[(size_t)k] = renumber[(size_t)code_k[(size_t)k]];
        }
      } catch (const std::bad_alloc&) {
        oom.store(true);
      }
    });
    if (oom.load()) return ffx_fail(FFX_OOM, "host allocation failed");
    std::vector<uint64_t>().swap(vals);
    std::vector<int64_t> base((size_t)S + 1, 0);
    for (int s = 0; s < S; ++s) base[(size_t)s + 1] = base[(size_t)s] + (int64_t)firsts[(size_t)s].size();
    next_shard.store(0);
    run_threads(T, [&](int) {
      for (;;) {
        const int s = next_shard.fetch_add(1);
        if (s >= S) return;
        const int32_t b = (int32_t)base[(size_t)s];
        for (int64_t k = shard_off[(size_t)s]; k < shard_off[(size_t)s + 1]; ++k) codes[rows[(size_t)k]] = code_k[(size_t)k] + b;
        int64_t w = base[(size_t)s];
        for (const int32_t i : firsts[(size_t)s]) first_row[w++] = i;
      }
    });
    *n_distinct = base[(size_t)S];
    return FFX_OK;
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
}

// out[off[r] .. off[r+1]) = r for r = 0 .. n_runs-1 (numpy.repeat(arange(n_runs), diff(off)) as int64), by host threads: the
// per-row query numbers of a run whose rows stand grouped by query -- 280 MB of fresh memory at 34.9 M rows, whose page faults
// one thread takes a second over and sixteen do not.
int ffx_host_fill_runs(const int64_t* off, int64_t n_runs, int32_t n_threads, int64_t* out) {
  if (n_runs < 0 || (n_runs > 0 && (!off || !out))) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n_runs == 0) return FFX_OK;
  if (off[0] != 0) return ffx_fail(FFX_BAD_ARG, "off[0] must be 0");
  for (int64_t r = 0; r < n_runs; ++r)
    if (off[r + 1] < off[r]) return ffx_fail(FFX_BAD_ARG, "off not monotone at %lld", (long long)r);
  const int64_t n = off[n_runs];
  int T = n_threads > 0 ? n_threads : usable_cores();
  T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)64, n / (1 << 18) + 1}));
  try {
    run_threads(T, [&](int t) {  // threads take ranges of ROWS (runs differ in length); the first run of a range by binary search
      const int64_t lo = n * t / T, hi = n * (t + 1) / T;
      if (lo >= hi) return;
      int64_t r = std::upper_bound(off, off + n_runs + 1, lo) - off - 1;
      for (int64_t i = lo; i < hi;) {
        while (off[r + 1] <= i) ++r;
        const int64_t end = std::min(hi, off[r + 1]);
        std::fill(out + i, out + end, r);
        i = end;
      }
    });
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
  return FFX_OK;
}

// Which rank of a sharded index holds row i: crc32 (the zlib / IEEE one, so that Python's zlib.crc32 names the same
// rank) of its document id -- or, for a row without one, of its passage id -- modulo `world`.  Ids arrive as the
// NUL-padded fixed-width fields an index file stores them in (numpy 'S<width>'; an empty field = no id).  Every rank
// of a ShardedOnDiskIndex runs this over the file's id columns block by block and interns only the ids of ITS rows: the
// string tables are sharded with the rows instead of being replicated (reference: OnDiskIndex.load reads every id into
// the one process there is, index/disk.py:355-418).  owner[i] = 255 for a row with neither id.
int ffx_host_owner_crc32(const char* doc_ids, int32_t doc_width, const char* psg_ids, int32_t psg_width, int64_t n, int32_t world, int32_t n_threads,
                         uint8_t* owner) {
  if (n < 0 || world < 1 || world > 254 || doc_width < 0 || psg_width < 0 || (n > 0 && (!owner || (!doc_ids && !psg_ids)))) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if ((doc_ids && doc_width < 1) || (psg_ids && psg_width < 1)) return ffx_fail(FFX_BAD_ARG, "a column of ids needs a width");
  static uint32_t table[256];
  static std::atomic<bool> ready(false);
  if (!ready.load(std::memory_order_acquire)) {  // (idempotent: concurrent first callers write the same values)
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
      table[i] = c;
    }
    ready.store(true, std::memory_order_release);
  }
  auto crc = [&](const char* p, size_t len) {
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < len; ++i) c = table[(c ^ (unsigned char)p[i]) & 0xFFu] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
  };
  int T = n_threads > 0 ? n_threads : usable_cores();
  T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)64, n / (1 << 16) + 1}));
  try {
    run_threads(T, [&](int t) {
      for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i) {
        const char* f = doc_ids ? doc_ids + (size_t)i * (size_t)doc_width : nullptr;
        size_t len = f ? strnlen(f, (size_t)doc_width) : 0;
        if (len == 0 && psg_ids) {
          f = psg_ids + (size_t)i * (size_t)psg_width;
          len = strnlen(f, (size_t)psg_width);
        }
        owner[i] = len == 0 ? (uint8_t)255 : (uint8_t)(crc(f, len) % (uint32_t)world);
      }
    });
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
  return FFX_OK;
}

int ffx_buffer_free(void* p) {
  free(p);
  return FFX_OK;
}

// every string of the table, in ordinal order, joined by `sep` (needed = bytes required; the copy
// happens only if cap >= needed)
int ffx_ids_blob(const ffx_ids* t, char sep, char* buf, int64_t cap, int64_t* needed) {
  if (!t || !needed) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  const int64_t n = t->size();
  const int64_t need = n == 0 ? 0 : (int64_t)t->arena.size() + (n - 1);
  *needed = need;
  if (!buf || cap < need) return FFX_OK;
  char* w = buf;
  for (int64_t i = 0; i < n; ++i) {
    const std::string_view s = t->get((int32_t)i);
    memcpy(w, s.data(), s.size());
    w += s.size();
    if (i + 1 < n) *w++ = sep;
  }
  return FFX_OK;
}

// out[i] = src[index[i]] for 4-byte elements (float32 scores, int32 id codes): what follows every device ordering
// on the host (reordering a run's columns by the per-query order, ranking.py:115-117's sort applied to the other
// columns).  NumPy does this on one core after widening the index to 64 bits; here threads take blocks of the
// output.  An index outside [0, n_src) is refused (nothing is written for it).
int ffx_host_take32(const void* src, int64_t n_src, const int32_t* index, int64_t n, int32_t n_threads, void* out) {
  if (n < 0 || n_src < 0) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n > 0 && (!src || !index || !out)) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  const uint32_t* s = static_cast<const uint32_t*>(src);
  uint32_t* o = static_cast<uint32_t*>(out);
  int T = n_threads > 0 ? n_threads : usable_cores();
  T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)64, n / (1 << 16) + 1}));
  std::atomic<int64_t> next(0), bad(-1);
  const int64_t block = 1 << 16;
  auto work = [&]() {
    for (;;) {
      const int64_t lo = next.fetch_add(block);
      if (lo >= n) return;
      const int64_t hi = std::min(n, lo + block);
      for (int64_t i = lo; i < hi; ++i) {
        const int64_t j = index[i];
        if ((uint64_t)j >= (uint64_t)n_src) {
          int64_t none = -1;
          bad.compare_exchange_strong(none, i);
          continue;
        }
        o[i] = s[j];
      }
    }
  };
  try {
    run_threads(T, [&](int) { work(); });
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
  if (bad.load() >= 0) return ffx_fail(FFX_BAD_ARG, "index[%lld] = %d is outside [0, %lld)", (long long)bad.load(), index[bad.load()], (long long)n_src);
  return FFX_OK;
}

// Ranking.interpolate (ranking.py:319) when the other ranking's rows are a known permutation of this one's
// (it came out of Index.__call__ on this ranking): out[j] = f32(a * sparse[j]) + f32(b * dense_of[i]) with
// j = source_row[i], i.e. the dense scores are put back in first-stage order and combined in one pass -- two
// rounded float32 multiplies and one add, never fused (this file is built with -ffp-contract=off), the
// arithmetic of NumPy's `alpha * s + (1 - alpha) * d` on float32 arrays and of the device's interp_kernel.
// source_row must be a permutation of 0..n-1 (checked; then every out element is written exactly once, which is
// what makes the threads independent); *any_nan reports a NaN among the results (the caller then takes the general path, which
// drops such rows as the reference does).
int ffx_host_interpolate(const float* sparse, const float* dense_of, const int32_t* source_row, int64_t n, float a, float b,
                         int32_t n_threads, float* out, int32_t* any_nan) {
  if (any_nan) *any_nan = 0;
  if (n < 0) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n > 0 && (!sparse || !dense_of || !source_row || !out)) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  if (n == 0) return FFX_OK;
  int T = n_threads > 0 ? n_threads : usable_cores();
  T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)64, n / (1 << 16) + 1}));
  uint8_t* seen = static_cast<uint8_t*>(calloc((size_t)n, 1));  // zero pages on first touch, no initialising pass
  if (!seen) return ffx_fail(FFX_OOM, "host allocation failed");
  std::atomic<int64_t> next(0), bad(-1), marked(0);
  std::atomic<int> nan(0);
  const int64_t block = 1 << 16;
  auto over_blocks = [&](auto&& body) -> bool {  // body(lo, hi) on blocks of 0..n, by T threads
    next.store(0);
    auto work = [&]() {
      for (;;) {
        const int64_t lo = next.fetch_add(block);
        if (lo >= n) return;
        body(lo, std::min(n, lo + block));
      }
    };
    try {
      run_threads(T, [&](int) { work(); });
    } catch (const std::system_error&) {
      return false;
    }
    return true;
  };
  const bool ran = over_blocks([&](int64_t lo, int64_t hi) {
    bool my_nan = false;
    for (int64_t i = lo; i < hi; ++i) {
      const int64_t j = source_row[i];
      if ((uint64_t)j >= (uint64_t)n) {
        int64_t none = -1;
        bad.compare_exchange_strong(none, i);
        continue;
      }
      __atomic_store_n(&seen[j], (uint8_t)1, __ATOMIC_RELAXED);  // a plain byte store (two threads may both mark a duplicate)
      const float x = a * sparse[j];
      const float y = b * dense_of[i];
      const float r = x + y;
      my_nan |= r != r;
      out[j] = r;
    }
    if (my_nan) nan.store(1);
  });
  // n entries, all inside 0..n-1: they are a permutation exactly when they marked n different rows
  const bool counted = ran && bad.load() < 0 && over_blocks([&](int64_t lo, int64_t hi) {
    int64_t c = 0;
    for (int64_t j = lo; j < hi; ++j) c += seen[j];
    marked.fetch_add(c);
  });
  free(seen);
  if (!ran || (bad.load() < 0 && !counted)) return ffx_fail(FFX_OOM, "could not start host threads");
  if (bad.load() >= 0) return ffx_fail(FFX_BAD_ARG, "source_row is not a permutation (entry %lld is outside 0..n-1)", (long long)bad.load());
  if (marked.load() != n)
    return ffx_fail(FFX_BAD_ARG, "source_row is not a permutation (%lld rows are named more than once)", (long long)(n - marked.load()));
  if (any_nan) *any_nan = nan.load();
  return FFX_OK;
}

// The distinct row numbers a batch gathers (each is read from the index file once, disk.py:309-336 reads them per
// batch too) in ascending order, and for every gathered row its position in that list: numpy.unique(rows,
// return_inverse=True) for values in 0 .. n_index_rows-1, without the sort -- a bitmap over the index's rows is marked by
// threads, ranked by a prefix over its 64-bit words, and read back.  uniq must hold min(n, n_index_rows) entries.
int ffx_host_unique_rows(const int64_t* rows, int64_t n, int64_t n_index_rows, int32_t n_threads, int64_t* uniq, int64_t* n_uniq, int32_t* inverse) {
  if (n_uniq) *n_uniq = 0;
  if (n < 0 || n_index_rows < 0 || !n_uniq || (n > 0 && (!rows || !uniq || !inverse))) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n == 0) return FFX_OK;
  if (n_index_rows > (int64_t)INT32_MAX) return ffx_fail(FFX_BAD_ARG, "too many index rows for int32 positions");
  int T = n_threads > 0 ? n_threads : usable_cores();
  T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)64, n / (1 << 16) + 1}));
  const int64_t n_words = (n_index_rows + 63) / 64;
  try {
    std::vector<std::atomic<uint64_t>> bits((size_t)std::max<int64_t>(n_words, 1));
    for (auto& w : bits) w.store(0, std::memory_order_relaxed);
    std::vector<int64_t> before((size_t)n_words + 1);
    std::atomic<int64_t> bad(-1);
    run_threads(T, [&](int t) {
      for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i) {
        const int64_t r = rows[i];
        if ((uint64_t)r >= (uint64_t)n_index_rows) {
          int64_t none = -1;
          bad.compare_exchange_strong(none, i);
          continue;
        }
        bits[(size_t)(r >> 6)].fetch_or(1ull << (r & 63), std::memory_order_relaxed);
      }
    });
    if (bad.load() >= 0) return ffx_fail(FFX_BAD_ARG, "rows[%lld] = %lld is outside the index", (long long)bad.load(), (long long)rows[bad.load()]);
    int64_t total = 0;
    for (int64_t w = 0; w < n_words; ++w) {
      before[(size_t)w] = total;
      total += __builtin_popcountll(bits[(size_t)w].load(std::memory_order_relaxed));
    }
    before[(size_t)n_words] = total;
    *n_uniq = total;
    run_threads(T, [&](int t) {
      for (int64_t w = n_words * t / T; w < n_words * (t + 1) / T; ++w) {
        uint64_t m = bits[(size_t)w].load(std::memory_order_relaxed);
        int64_t at = before[(size_t)w];
        while (m) {
          uniq[at++] = (w << 6) + __builtin_ctzll(m);
          m &= m - 1;
        }
      }
    });
    run_threads(T, [&](int t) {
      for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i) {
        const int64_t r = rows[i];
        const uint64_t m = bits[(size_t)(r >> 6)].load(std::memory_order_relaxed);
        inverse[i] = (int32_t)(before[(size_t)(r >> 6)] + __builtin_popcountll(m & ((1ull << (r & 63)) - 1ull)));
      }
    });
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
  return FFX_OK;
}

// get_indices (index/util.py:29-41) for candidates that contribute ONE row each, and numpy.unique of those rows, in one go:
//   row(i) = table2 ? table2[table1[ids[i]]] : table1[ids[i]]
// (PASSAGE: table1 = the passages' rows; FIRSTP / one-row documents: table1 = the documents' first entry, table2 = the
// entries' rows), then what ffx_host_unique_rows returns for them.  A streamed index does this once per call before its
// first window can start: two thread pools instead of five, no intermediate arrays.
int ffx_host_rows_unique(const int32_t* ids, int64_t n, const int32_t* table1, int64_t n1, const int32_t* table2, int64_t n2,
                         int64_t n_index_rows, int32_t n_threads, int64_t* uniq, int64_t* n_uniq, int32_t* inverse) {
  if (n_uniq) *n_uniq = 0;
  if (n < 0 || n1 < 0 || n2 < 0 || n_index_rows < 0 || !n_uniq || (n > 0 && (!ids || !table1 || !uniq || !inverse))) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n == 0) return FFX_OK;
  if (n_index_rows > (int64_t)INT32_MAX) return ffx_fail(FFX_BAD_ARG, "too many index rows for int32 positions");
  int T = n_threads > 0 ? n_threads : usable_cores();
  T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)64, n / (1 << 16) + 1}));
  const int64_t n_words = (n_index_rows + 63) / 64;
  try {
    std::vector<std::atomic<uint64_t>> bits((size_t)std::max<int64_t>(n_words, 1));
    for (auto& w : bits) w.store(0, std::memory_order_relaxed);
    std::vector<int64_t> before((size_t)n_words + 1);
    std::unique_ptr<int32_t[]> row_of(new int32_t[(size_t)n]);
    std::atomic<int64_t> bad(-1);
    run_threads(T, [&](int t) {
      // block by block: first the look-ups alone (independent loads, many in flight), then the bitmap -- an atomic
      // read-modify-write between two look-ups would make every one of them wait for the one before
      const int64_t lo = n * t / T, hi = n * (t + 1) / T;
      for (int64_t b0 = lo; b0 < hi; b0 += 4096) {
        const int64_t b1 = std::min(hi, b0 + 4096);
        for (int64_t i = b0; i < b1; ++i) {  // one table per loop: one cache miss per iteration, none waiting for another
          const int64_t r = ids[i];
          row_of[(size_t)i] = (uint64_t)r < (uint64_t)n1 ? table1[r] : -1;
        }
        if (table2)
          for (int64_t i = b0; i < b1; ++i) {
            const int64_t r = row_of[(size_t)i];
            row_of[(size_t)i] = (uint64_t)r < (uint64_t)n2 ? table2[r] : -1;
          }
        for (int64_t i = b0; i < b1; ++i)
          if ((uint64_t)(int64_t)row_of[(size_t)i] >= (uint64_t)n_index_rows) {
            int64_t none = -1;
            bad.compare_exchange_strong(none, i);
            row_of[(size_t)i] = -1;
          }
        for (int64_t i = b0; i < b1; ++i) {
          const int32_t r = row_of[(size_t)i];
          if (r >= 0) bits[(size_t)(r >> 6)].fetch_or(1ull << (r & 63), std::memory_order_relaxed);
        }
      }
    });
    if (bad.load() >= 0) return ffx_fail(FFX_BAD_ARG, "ids[%lld] = %d names no row of the index", (long long)bad.load(), ids[bad.load()]);
    int64_t total = 0;
    for (int64_t w = 0; w < n_words; ++w) {
      before[(size_t)w] = total;
      total += __builtin_popcountll(bits[(size_t)w].load(std::memory_order_relaxed));
    }
    before[(size_t)n_words] = total;
    *n_uniq = total;
    run_threads(T, [&](int t) {
      for (int64_t w = n_words * t / T; w < n_words * (t + 1) / T; ++w) {
        uint64_t m = bits[(size_t)w].load(std::memory_order_relaxed);
        int64_t at = before[(size_t)w];
        while (m) {
          uniq[at++] = (w << 6) + __builtin_ctzll(m);
          m &= m - 1;
        }
      }
      for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i) {
        const int64_t r = row_of[(size_t)i];
        const uint64_t m = bits[(size_t)(r >> 6)].load(std::memory_order_relaxed);
        inverse[i] = (int32_t)(before[(size_t)(r >> 6)] + __builtin_popcountll(m & ((1ull << (r & 63)) - 1ull)));
      }
    });
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
  return FFX_OK;
}

// get_indices (index/util.py:29-41) for candidates that contribute ALL rows of their document (MaxP, AveP over multi-passage
// documents): entry_off[id] .. entry_off[id + 1] are the document's entries, entry_row[] their rows.  Two calls, because the
// caller allocates by the total: ffx_host_segment_offsets gives seg_off[0 .. n] (seg_off[n] = R gathered rows),
// ffx_host_segment_rows_unique then the distinct rows ascending and the slot of every gathered row among them.
int ffx_host_segment_offsets(const int32_t* ids, int64_t n, const int32_t* entry_off, int64_t n_docs, int32_t n_threads, int64_t* seg_off) {
  if (n < 0 || n_docs < 0 || !seg_off || (n > 0 && (!ids || !entry_off))) return ffx_fail(FFX_BAD_ARG, "bad argument");
  seg_off[0] = 0;
  if (n == 0) return FFX_OK;
  int T = n_threads > 0 ? n_threads : usable_cores();
  T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)64, n / (1 << 16) + 1}));
  std::vector<int64_t> part((size_t)T + 1, 0);
  std::atomic<int64_t> bad(-1);
  try {
    run_threads(T, [&](int t) {  // counts, and their running sum inside the thread's block
      int64_t sum = 0;
      for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i) {
        const int64_t d = ids[i];
        if ((uint64_t)d >= (uint64_t)n_docs || entry_off[d + 1] < entry_off[d]) {
          int64_t none = -1;
          bad.compare_exchange_strong(none, i);
          seg_off[i + 1] = sum;
          continue;
        }
        sum += entry_off[d + 1] - entry_off[d];
        seg_off[i + 1] = sum;
      }
      part[(size_t)t + 1] = sum;
    });
    if (bad.load() >= 0) return ffx_fail(FFX_BAD_ARG, "ids[%lld] = %d names no document of the index", (long long)bad.load(), ids[bad.load()]);
    for (int t = 0; t < T; ++t) part[(size_t)t + 1] += part[(size_t)t];
    run_threads(T, [&](int t) {
      const int64_t add = part[(size_t)t];
      if (add)
        for (int64_t i = n * t / T; i < n * (t + 1) / T; ++i) seg_off[i + 1] += add;
    });
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
  return FFX_OK;
}

int ffx_host_segment_rows_unique(const int32_t* ids, int64_t n, const int32_t* entry_off, int64_t n_docs, const int32_t* entry_row,
                                 int64_t n_entries, const int64_t* seg_off, int64_t n_index_rows, int32_t n_threads, int64_t* uniq,
                                 int64_t* n_uniq, int32_t* inverse) {
  if (n_uniq) *n_uniq = 0;
  if (n < 0 || n_docs < 0 || n_entries < 0 || n_index_rows < 0 || !n_uniq || !seg_off || (n > 0 && (!ids || !entry_off || !entry_row || !uniq || !inverse)))
    return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (n == 0 || seg_off[n] == 0) return FFX_OK;
  if (n_index_rows > (int64_t)INT32_MAX) return ffx_fail(FFX_BAD_ARG, "too many index rows for int32 positions");
  const int64_t R = seg_off[n];
  int T = n_threads > 0 ? n_threads : usable_cores();
  T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)64, R / (1 << 16) + 1}));
  const int64_t n_words = (n_index_rows + 63) / 64;
  try {
    std::vector<uint64_t> bits((size_t)std::max<int64_t>(n_words, 1), 0);
    std::vector<int64_t> before((size_t)n_words + 1);
    std::atomic<int64_t> bad(-1);
    // threads take blocks of candidates of about equal numbers of gathered rows; `inverse` holds the rows themselves first
    auto first_pair = [&](int t) { return (int64_t)(std::lower_bound(seg_off, seg_off + n + 1, R * t / T) - seg_off); };
    run_threads(T, [&](int t) {
      const int64_t lo = t == 0 ? 0 : std::min<int64_t>(n, first_pair(t)), hi = t + 1 == T ? n : std::min<int64_t>(n, first_pair(t + 1));
      for (int64_t i = lo; i < hi; ++i) {
        const int64_t d = ids[i];
        const int64_t len = seg_off[i + 1] - seg_off[i];
        if ((uint64_t)d >= (uint64_t)n_docs || entry_off[d + 1] - entry_off[d] != len || entry_off[d] < 0 || entry_off[d] + len > n_entries) {
          int64_t none = -1;
          bad.compare_exchange_strong(none, i);
          for (int64_t k = 0; k < len; ++k) inverse[seg_off[i] + k] = 0;
          continue;
        }
        const int32_t* src = entry_row + entry_off[d];
        for (int64_t k = 0; k < len; ++k) {
          const int32_t r = src[k];
          if ((uint64_t)(int64_t)r >= (uint64_t)n_index_rows) {
            int64_t none = -1;
            bad.compare_exchange_strong(none, i);
            inverse[seg_off[i] + k] = 0;
          } else {
            inverse[seg_off[i] + k] = r;
          }
        }
      }
    });
    if (bad.load() >= 0) return ffx_fail(FFX_BAD_ARG, "ids[%lld] = %d: its rows do not match the offsets, or lie outside the index", (long long)bad.load(), ids[bad.load()]);
    mark_rows(bits.data(), inverse, R, T);
    int64_t total = 0;
    for (int64_t w = 0; w < n_words; ++w) {
      before[(size_t)w] = total;
      total += __builtin_popcountll(bits[(size_t)w]);
    }
    before[(size_t)n_words] = total;
    *n_uniq = total;
    run_threads(T, [&](int t) {
      for (int64_t w = n_words * t / T; w < n_words * (t + 1) / T; ++w) {
        uint64_t m = bits[(size_t)w];
        int64_t at = before[(size_t)w];
        while (m) {
          uniq[at++] = (w << 6) + __builtin_ctzll(m);
          m &= m - 1;
        }
      }
      for (int64_t p = R * t / T; p < R * (t + 1) / T; ++p) {
        const int64_t r = inverse[p];
        const uint64_t m = bits[(size_t)(r >> 6)];
        inverse[p] = (int32_t)(before[(size_t)(r >> 6)] + __builtin_popcountll(m & ((1ull << (r & 63)) - 1ull)));
      }
    });
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
  return FFX_OK;
}

// The outer join of Ranking.interpolate / __add__ (ranking.py:293-326: merge on (q_id, id), how="outer") for two
// rankings whose rows stand grouped by query, on integer keys: query g owns rows off_a[g] .. off_a[g+1] of A and rows
// off_b[g] .. off_b[g+1] of B; code_a / code_b number the ids in one common table; a (query, id) pair occurs at most
// once per side.  For every query the union of the two id sets, in ascending id code:
//   u_code[i]  the id code          u_a[i] / u_b[i]  the row of A / B holding the pair, -1 where that side lacks it
// and u_off[g] .. u_off[g+1] the union rows of query g.  The outputs hold off_a[n] + off_b[n] entries (the worst case);
// u_off[n_queries] says how many are used.  Threads take blocks of queries; every query is sorted by id code on both
// sides (64-bit words of code and row) and merged, once to count and once to write.
int ffx_host_outer_join(const int64_t* off_a, const int32_t* code_a, const int64_t* off_b, const int32_t* code_b, int64_t n_queries,
                        int32_t n_threads, int64_t* u_off, int32_t* u_code, int32_t* u_a, int32_t* u_b) {
  if (!off_a || !off_b || !u_off || n_queries < 0) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (off_a[0] != 0 || off_b[0] != 0) return ffx_fail(FFX_BAD_ARG, "offsets must start at 0");
  for (int64_t g = 0; g < n_queries; ++g)
    if (off_a[g + 1] < off_a[g] || off_b[g + 1] < off_b[g]) return ffx_fail(FFX_BAD_ARG, "offsets not monotone at query %lld", (long long)g);
  const int64_t na = off_a[n_queries], nb = off_b[n_queries];
  if (na > INT32_MAX - 1 || nb > INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "too many rows");
  if ((na > 0 && !code_a) || (nb > 0 && !code_b) || ((na + nb) > 0 && (!u_code || !u_a || !u_b))) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  int T = n_threads > 0 ? n_threads : usable_cores();
  T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)64, (na + nb) / (1 << 16) + 1, std::max<int64_t>(n_queries, 1)}));
  try {
    std::vector<uint64_t> word_a((size_t)na), word_b((size_t)nb);  // (code << 32 | row), sorted inside every query
    std::atomic<int64_t> next(0);
    std::atomic<int64_t> dup(-1);
    u_off[0] = 0;
    run_threads(T, [&](int) {
      for (;;) {
        const int64_t g0 = next.fetch_add(8);
        if (g0 >= n_queries) return;
        for (int64_t g = g0; g < std::min(n_queries, g0 + 8); ++g) {
          for (int64_t i = off_a[g]; i < off_a[g + 1]; ++i) word_a[(size_t)i] = ((uint64_t)(uint32_t)code_a[i] << 32) | (uint32_t)i;
          for (int64_t i = off_b[g]; i < off_b[g + 1]; ++i) word_b[(size_t)i] = ((uint64_t)(uint32_t)code_b[i] << 32) | (uint32_t)i;
          std::sort(word_a.begin() + off_a[g], word_a.begin() + off_a[g + 1]);
          std::sort(word_b.begin() + off_b[g], word_b.begin() + off_b[g + 1]);
          int64_t i = off_a[g], j = off_b[g], n = 0;
          const int64_t ie = off_a[g + 1], je = off_b[g + 1];
          while (i < ie && j < je) {
            const uint32_t ca = (uint32_t)(word_a[(size_t)i] >> 32), cb = (uint32_t)(word_b[(size_t)j] >> 32);
            i += ca <= cb;
            j += cb <= ca;
            ++n;
          }
          n += (ie - i) + (je - j);
          for (int64_t x = off_a[g] + 1; x < ie; ++x)
            if ((word_a[(size_t)x] >> 32) == (word_a[(size_t)x - 1] >> 32)) {
              int64_t none = -1;
              dup.compare_exchange_strong(none, g);
            }
          for (int64_t x = off_b[g] + 1; x < je; ++x)
            if ((word_b[(size_t)x] >> 32) == (word_b[(size_t)x - 1] >> 32)) {
              int64_t none = -1;
              dup.compare_exchange_strong(none, g);
            }
          u_off[g + 1] = n;  // counts now, offsets after the scan below
        }
      }
    });
    if (dup.load() >= 0) return ffx_fail(FFX_BAD_ARG, "query %lld holds an id twice on one side", (long long)dup.load());
    for (int64_t g = 0; g < n_queries; ++g) u_off[g + 1] += u_off[g];
    next.store(0);
    run_threads(T, [&](int) {
      for (;;) {
        const int64_t g0 = next.fetch_add(8);
        if (g0 >= n_queries) return;
        for (int64_t g = g0; g < std::min(n_queries, g0 + 8); ++g) {
          int64_t i = off_a[g], j = off_b[g], at = u_off[g];
          const int64_t ie = off_a[g + 1], je = off_b[g + 1];
          while (i < ie || j < je) {
            const uint64_t wa = i < ie ? word_a[(size_t)i] : ~0ull, wb = j < je ? word_b[(size_t)j] : ~0ull;
            const uint32_t ca = (uint32_t)(wa >> 32), cb = (uint32_t)(wb >> 32);
            const bool take_a = i < ie && (j >= je || ca <= cb), take_b = j < je && (i >= ie || cb <= ca);
            u_code[at] = (int32_t)(take_a ? ca : cb);
            u_a[at] = take_a ? (int32_t)(uint32_t)wa : -1;
            u_b[at] = take_b ? (int32_t)(uint32_t)wb : -1;
            i += take_a;
            j += take_b;
            ++at;
          }
        }
      }
    });
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
  return FFX_OK;
}

// Per-query order on the host: scores descending, exact ties by position (stable), -0.0 counted as
// +0.0 -- the order FFX_FLAG_SORT produces on the device (ranking.py:115-117), for score arrays that were
// combined on the host (Ranking.interpolate).  Threads take blocks of queries; out_order[p] is a
// position inside the whole array, as ffx_plan_fetch hands it out.
int ffx_host_order(const float* scores, const int64_t* pair_off, int64_t n_queries, int32_t n_threads, int32_t* out_order) {
  return ffx_host_order_blocks(scores, pair_off, nullptr, n_queries, n_threads, out_order);
}

// ffx_host_order with the queries' blocks laid out anew: the order of query q (positions in the input, as before) is
// written from out_order[dst_off[q]] on -- ordering every query AND putting the queries into another sequence (the
// constructor's "q_id descending") in the one pass.  dst_off NULL: where the query stands (ffx_host_order).  The
// caller sees to it that the destination blocks do not overlap.
int ffx_host_order_blocks(const float* scores, const int64_t* pair_off, const int64_t* dst_off, int64_t n_queries, int32_t n_threads,
                          int32_t* out_order) {
  return ffx_host_rank_blocks(scores, nullptr, pair_off, dst_off, n_queries, n_threads, out_order, nullptr, nullptr);
}

// ffx_host_order_blocks with the columns moved along in the same pass: out_scores / out_carry receive the scores and a
// second 4-byte column (the id codes) of every query in its final order at the query's destination block -- while the
// query's 5000 words are in the core's cache, instead of two gathers over the whole run afterwards.  The constructor's sort
// (ranking.py:115-117) of a run that stands grouped by query, in one pass.  Any output may be NULL.
int ffx_host_rank_blocks(const float* scores, const int32_t* carry, const int64_t* pair_off, const int64_t* dst_off, int64_t n_queries,
                         int32_t n_threads, int32_t* out_order, float* out_scores, int32_t* out_carry) {
  if (!pair_off || n_queries < 0 || pair_off[0] != 0) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (dst_off)
    for (int64_t q = 0; q < n_queries; ++q)
      if (dst_off[q] < 0 || dst_off[q] + (pair_off[q + 1] - pair_off[q]) > pair_off[n_queries]) return ffx_fail(FFX_BAD_ARG, "destination of query %lld lies outside the output", (long long)q);
  for (int64_t q = 0; q < n_queries; ++q)
    if (pair_off[q + 1] < pair_off[q]) return ffx_fail(FFX_BAD_ARG, "pair_off not monotone at query %lld", (long long)q);
  const int64_t P = pair_off[n_queries];
  if (P > INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "too many pairs");
  if (P > 0 && (!scores || (out_carry && !carry))) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  auto image = [](float x) -> uint32_t {  // larger score <=> larger image
    uint32_t b;
    memcpy(&b, &x, sizeof b);
    if (b == 0x80000000u) b = 0u;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  };
  int T = n_threads > 0 ? n_threads : usable_cores();
  T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)64, P / (1 << 16) + 1, std::max<int64_t>(n_queries, 1)}));
  std::atomic<int64_t> next(0);
  std::atomic<bool> oom(false);
  auto work = [&]() {
    try {
      std::vector<uint64_t> keys, other;
      for (;;) {
        const int64_t q0 = next.fetch_add(16);
        if (q0 >= n_queries) return;
        for (int64_t q = q0; q < std::min(n_queries, q0 + 16); ++q) {
          const int64_t lo = pair_off[q], n = pair_off[q + 1] - lo;
          keys.resize((size_t)n);
          // one 64-bit word per candidate: (inverted image, position) ascending = score descending, ties by position
          for (int64_t i = 0; i < n; ++i) keys[(size_t)i] = ((uint64_t)(~image(scores[lo + i])) << 32) | (uint32_t)i;
          if (n < 256) {
            std::sort(keys.begin(), keys.end());
          } else {
            // least-significant-digit radix sort on the four bytes of the image: every pass is a stable counting
            // sort and the words start in position order, so equal images stay in position order (what sorting
            // the whole word gives); a byte all candidates share (typically the exponent's) costs only its count.
            // A query's 5000 words stay in L1/L2 throughout: ~4x faster than std::sort at that size.
            other.resize((size_t)n);
            uint32_t count[4][256];
            memset(count, 0, sizeof count);
            for (int64_t i = 0; i < n; ++i) {
              const uint32_t k = (uint32_t)(keys[(size_t)i] >> 32);
              ++count[0][k & 255];
              ++count[1][(k >> 8) & 255];
              ++count[2][(k >> 16) & 255];
              ++count[3][k >> 24];
            }
            uint64_t* from = keys.data();
            uint64_t* to = other.data();
            for (int pass = 0; pass < 4; ++pass) {
              uint32_t* c = count[pass];
              const int shift = 32 + 8 * pass;
              if (c[(from[0] >> shift) & 255] == (uint32_t)n) continue;  // one bucket holds everything
              uint32_t at = 0;
              for (int b = 0; b < 256; ++b) {
                const uint32_t here = c[b];
                c[b] = at;
                at += here;
              }
              for (int64_t i = 0; i < n; ++i) to[c[(from[i] >> shift) & 255]++] = from[i];
              std::swap(from, to);
            }
            if (from != keys.data()) keys.swap(other);
          }
          const int64_t at = dst_off ? dst_off[q] : lo;
          if (out_order)
            for (int64_t i = 0; i < n; ++i) out_order[at + i] = (int32_t)(lo + (int64_t)(keys[(size_t)i] & 0xFFFFFFFFu));
          if (out_scores)
            for (int64_t i = 0; i < n; ++i) out_scores[at + i] = scores[lo + (int64_t)(keys[(size_t)i] & 0xFFFFFFFFu)];
          if (out_carry)
            for (int64_t i = 0; i < n; ++i) out_carry[at + i] = carry[lo + (int64_t)(keys[(size_t)i] & 0xFFFFFFFFu)];
        }
      }
    } catch (const std::bad_alloc&) {
      oom.store(true);
    }
  };
  try {
    run_threads(T, [&](int) { work(); });  // (threads that did start are joined before a start-up error gets here)
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
  return oom.load() ? ffx_fail(FFX_OOM, "host allocation failed") : FFX_OK;
}

int ffx_ids_get(const ffx_ids* t, int32_t ordinal, const char** str, int32_t* len) {
  if (!t || !str || !len) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  if (ordinal < 0 || ordinal >= t->size()) return ffx_fail(FFX_BAD_ARG, "ordinal %d out of range", ordinal);
  const std::string_view s = t->get(ordinal);
  *str = s.data();
  *len = (int32_t)s.size();
  return FFX_OK;
}

}  // extern "C"

extern "C" int ffx_host_stage_report(char* buf, int64_t cap, int64_t* needed) {
  if (!needed) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  const std::string& r = stage_report();
  *needed = (int64_t)r.size() + 1;
  if (buf && cap >= *needed) memcpy(buf, r.c_str(), r.size() + 1);
  return FFX_OK;
}
