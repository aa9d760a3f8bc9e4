// string_code.h -- numbering the strings of a column without keeping a table (the factorisation of a run's id column,
// src/fast_forward/ranking.py:95-98 / pandas.factorize), by host threads.  Shared by host_io.cpp (ffx_strings_code*: joined
// buffers, Arrow offsets) and run_file.cpp (the candidate ids of a parsed run file).
//
// codes[i] = number of string i; equal strings get equal numbers; the numbers are dense and ascend in (name_hash, bytes), so
// they depend on the SET of strings only -- not on the order of the rows, the thread count or the process: ranks of a sharded
// index number the same ids alike and can talk about them by number.
//
// How: every row becomes one 16-byte record (hash, row, length) and the records are dealt into 1024 shards by the upper hash
// bits (ranges of the hash, so shard order is number order).  A shard is numbered on its own: its records are read front to
// back, its table (open addressing) and the records of its few thousand distinct strings stay in the core's cache, and an id
// of up to eight bytes is compared through its hash alone (name_hash is a bijection on those) -- the text it came from is not
// touched again until the distinct names are written.  Fresh memory: 16 bytes per row (page faults are what large host
// passes wait for).
#ifndef FFX_STRING_CODE_H_
#define FFX_STRING_CODE_H_

#include <atomic>

#include "host_util.h"

namespace ffx_host {

struct CodeRec {
  uint64_t hash;
  uint32_t row;
  uint32_t len;  // the string's length; once its shard is numbered: the shard-local number of the string
};

constexpr int kCodeShards = 1024;
constexpr int kCodeShardShift = 54;  // shard = hash >> 54

// view(i) -> std::string_view of string i, valid until this returns.  *names_blob is malloc'ed (ffx_buffer_free) and holds
// the distinct strings in number order joined by sep, *names_bytes without a trailing separator.
template <class View>
int strings_code_all(int64_t n, int T, View&& view, char sep, int32_t* codes, char** names_blob, int64_t* names_bytes, int64_t* n_names,
                     StageClock* clock = nullptr, Buf<int64_t>* name_start = nullptr) {
  constexpr int S = kCodeShards;
  *names_blob = nullptr;
  *names_bytes = 0;
  *n_names = 0;
  if (n == 0) {
    if (name_start) {
      name_start->alloc(1);
      (*name_start)[0] = 1;
    }
    return FFX_OK;
  }
  auto row_range = [&](int t) { return std::pair<int64_t, int64_t>(n * t / T, n * (t + 1) / T); };
  // 1. how many rows of every row range fall into every shard
  std::vector<int64_t> count((size_t)T * S, 0);
  std::atomic<bool> too_long(false);
  run_threads(T, [&](int t) {
    const auto [lo, hi] = row_range(t);
    int64_t* c = count.data() + (size_t)t * S;
    for (int64_t i = lo; i < hi; ++i) {
      const std::string_view v = view(i);
      if (v.size() > (size_t)INT32_MAX) {
        too_long.store(true);
        return;
      }
      ++c[name_hash(v.data(), v.size()) >> kCodeShardShift];
    }
  });
  if (too_long.load()) return ffx_fail(FFX_BAD_ARG, "a string of more than 2 GiB cannot be numbered");
  if (clock) clock->lap("number_ids.count");
  // 2. the records dealt into shards, row order kept inside a shard
  std::vector<int64_t> shard_off((size_t)S + 1, 0);
  {
    int64_t at = 0;
    for (int s = 0; s < S; ++s) {
      shard_off[(size_t)s] = at;
      for (int t = 0; t < T; ++t) {
        const int64_t c = count[(size_t)t * S + s];
        count[(size_t)t * S + s] = at;  // becomes: where range t writes its rows of shard s
        at += c;
      }
    }
    shard_off[(size_t)S] = at;
  }
  Buf<CodeRec> recs((size_t)n);
  run_threads(T, [&](int t) {
    const auto [lo, hi] = row_range(t);
    int64_t* at = count.data() + (size_t)t * S;
    for (int64_t i = lo; i < hi; ++i) {
      const std::string_view v = view(i);
      const uint64_t h = name_hash(v.data(), v.size());
      recs[(size_t)at[h >> kCodeShardShift]++] = CodeRec{h, (uint32_t)i, (uint32_t)v.size()};
    }
  });
  if (clock) clock->lap("number_ids.deal");
  // 3. every shard numbered on its own, in (hash, bytes) order; afterwards rec.len holds the shard-local number
  auto same = [&](const CodeRec& a, const CodeRec& b) {
    if (a.hash != b.hash || a.len != b.len) return false;
    return a.len <= 8 || memcmp(view(a.row).data(), view(b.row).data(), a.len) == 0;
  };
  std::vector<std::vector<int32_t>> firsts((size_t)S);  // per shard: a row of every distinct string, in number order
  std::vector<int64_t> shard_bytes((size_t)S, 0);
  std::atomic<int> next_shard(0);
  std::atomic<bool> oom(false);
  run_threads(T, [&](int) {
    try {
      std::vector<int32_t> slot, by_content, renumber;
      std::vector<CodeRec> distinct;  // (compact: the probes compare against these, not against records all over the shard)
      for (;;) {
        const int s = next_shard.fetch_add(1);
        if (s >= S) return;
        const int64_t lo = shard_off[(size_t)s], hi = shard_off[(size_t)s + 1];
        if (lo == hi) continue;
        CodeRec* rec = recs.data() + lo;
        const int64_t m = hi - lo;
        size_t cap = 64;
        slot.assign(cap, -1);  // slot -> shard-local number (in order of first appearance, for now)
        distinct.clear();
        for (int64_t k = 0; k < m; ++k) {
          CodeRec& r = rec[k];
          size_t at = (size_t)(r.hash >> 12) & (cap - 1);
          for (;;) {
            const int32_t have = slot[at];
            if (have < 0) {
              slot[at] = (int32_t)distinct.size();
              distinct.push_back(r);
              r.len = (uint32_t)(distinct.size() - 1);
              break;
            }
            if (same(distinct[(size_t)have], r)) {
              r.len = (uint32_t)have;
              break;
            }
            at = (at + 1) & (cap - 1);
          }
          if (distinct.size() * 2 > cap) {  // grow: the table follows the number of DISTINCT strings and stays in cache
            cap *= 4;
            slot.assign(cap, -1);
            for (size_t j = 0; j < distinct.size(); ++j) {
              size_t p = (size_t)(distinct[j].hash >> 12) & (cap - 1);
              while (slot[p] >= 0) p = (p + 1) & (cap - 1);
              slot[p] = (int32_t)j;
            }
          }
        }
        const size_t d = distinct.size();
        by_content.resize(d);
        renumber.resize(d);
        for (size_t j = 0; j < d; ++j) by_content[j] = (int32_t)j;
        std::sort(by_content.begin(), by_content.end(), [&](int32_t x, int32_t y) {
          const CodeRec &a = distinct[(size_t)x], &b = distinct[(size_t)y];
          return a.hash != b.hash ? a.hash < b.hash : view(a.row) < view(b.row);
        });
        auto& first = firsts[(size_t)s];
        first.resize(d);
        int64_t bytes = 0;
        for (size_t j = 0; j < d; ++j) {
          const CodeRec& a = distinct[(size_t)by_content[j]];
          renumber[(size_t)by_content[j]] = (int32_t)j;
          first[j] = (int32_t)a.row;
          bytes += a.len;
        }
        for (int64_t k = 0; k < m; ++k) rec[k].len = (uint32_t)renumber[rec[k].len];
        shard_bytes[(size_t)s] = bytes;
      }
    } catch (const std::bad_alloc&) {  // (an exception leaving a thread would end the process)
      oom.store(true);
    }
  });
  if (oom.load()) return ffx_fail(FFX_OOM, "host allocation failed");
  if (clock) clock->lap("number_ids.shards");
  // 4. shard-local numbers -> global numbers; the names written shard by shard
  std::vector<int64_t> base((size_t)S + 1, 0), at_byte((size_t)S + 1, 0);
  for (int s = 0; s < S; ++s) {
    base[(size_t)s + 1] = base[(size_t)s] + (int64_t)firsts[(size_t)s].size();
    at_byte[(size_t)s + 1] = at_byte[(size_t)s] + shard_bytes[(size_t)s] + (int64_t)firsts[(size_t)s].size();  // + one separator each
  }
  const int64_t total = at_byte[(size_t)S];  // one separator too many (after the last name)
  char* blob = static_cast<char*>(malloc((size_t)std::max<int64_t>(total, 1)));
  if (!blob) return ffx_fail(FFX_OOM, "host allocation failed");
  if (name_start) {  // name c is blob[start[c] .. start[c+1] - 1)
    try {
      name_start->alloc((size_t)base[(size_t)S] + 1);
    } catch (const std::bad_alloc&) {
      free(blob);
      return ffx_fail(FFX_OOM, "host allocation failed");
    }
    (*name_start)[(size_t)base[(size_t)S]] = total;
  }
  next_shard.store(0);
  run_threads(T, [&](int) {
    for (;;) {
      const int s = next_shard.fetch_add(1);
      if (s >= S) return;
      const int64_t lo = shard_off[(size_t)s], hi = shard_off[(size_t)s + 1];
      const CodeRec* rec = recs.data() + lo;
      const int32_t b = (int32_t)base[(size_t)s];
      for (int64_t k = 0; k < hi - lo; ++k) codes[rec[k].row] = (int32_t)rec[k].len + b;
      char* w = blob + at_byte[(size_t)s];
      int64_t number = base[(size_t)s];
      for (const int32_t row : firsts[(size_t)s]) {
        const std::string_view v = view(row);
        if (name_start) (*name_start)[(size_t)number++] = w - blob;
        memcpy(w, v.data(), v.size());
        w += v.size();
        *w++ = sep;
      }
    }
  });
  *names_blob = blob;
  *names_bytes = total - 1;
  *n_names = base[(size_t)S];
  if (clock) clock->lap("number_ids.codes_and_names");
  return FFX_OK;
}

// The entry point.  A column of long RUNS of one string -- the query-id column of a run names each query k rows in a row -- is
// numbered through its run heads: one streaming pass compares neighbours, the few heads are numbered by the general scheme
// (the numbers depend on the set of strings only, so they are the numbers the whole column would get) and every row takes its
// head's number.  Anything else: the general scheme over all rows.
template <class View>
int strings_code_views(int64_t n, int T, View&& view, char sep, int32_t* codes, char** names_blob, int64_t* names_bytes, int64_t* n_names,
                       StageClock* clock = nullptr, Buf<int64_t>* name_start = nullptr) {
  if (n >= (1 << 16)) {
    std::vector<std::vector<int64_t>> heads_of((size_t)T);
    std::atomic<bool> oom(false);
    run_threads(T, [&](int t) {
      try {
        auto& mine = heads_of[(size_t)t];
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        for (int64_t i = lo; i < hi; ++i)
          if (i == 0 || view(i) != view(i - 1)) {
            if ((int64_t)mine.size() > (hi - lo) / 64 + 16) return;  // no long runs here: give up early
            mine.push_back(i);
          }
      } catch (const std::bad_alloc&) {
        oom.store(true);
      }
    });
    if (oom.load()) return ffx_fail(FFX_OOM, "host allocation failed");
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
      std::vector<int32_t> head_code((size_t)H);
      const int rc = strings_code_all(H, threads_for(H, 1 << 16, T), [&](int64_t j) { return view(head[(size_t)j]); }, sep, head_code.data(), names_blob,
                                      names_bytes, n_names, nullptr, name_start);
      if (rc != FFX_OK) return rc;
      run_threads(T, [&](int t) {
        for (int64_t h = H * t / T; h < H * (t + 1) / T; ++h) std::fill(codes + head[(size_t)h], codes + head[(size_t)h + 1], head_code[(size_t)h]);
      });
      return FFX_OK;
    }
    if (clock) clock->lap("number_ids.look_for_runs");
  }
  return strings_code_all(n, T, view, sep, codes, names_blob, names_bytes, n_names, clock, name_start);
}

// The first group of rows (group_off[g] .. group_off[g+1]) in which one code occurs twice, -1 if there is none: the repeated
// (query, id) pair of ranking.py:95-98 once the ids are numbers.  One mark per code and thread, set and cleared group by
// group.  Codes lie in [0, n_codes).
inline int64_t first_group_with_repeated_code(const int32_t* codes, const int64_t* group_off, int64_t n_groups, int64_t n_codes, int T) {
  std::atomic<int64_t> found{INT64_MAX};
  std::atomic<bool> oom(false);
  std::atomic<int64_t> next(0);
  run_threads(T, [&](int) {
    std::vector<uint8_t> seen;
    try {
      seen.assign((size_t)n_codes, 0);
    } catch (const std::bad_alloc&) {
      oom.store(true);
      return;
    }
    for (;;) {
      const int64_t g0 = next.fetch_add(16);
      if (g0 >= n_groups || g0 > found.load(std::memory_order_relaxed)) return;
      for (int64_t g = g0; g < std::min(n_groups, g0 + 16); ++g) {
        const int64_t lo = group_off[g], hi = group_off[g + 1];
        bool dup = false;
        for (int64_t i = lo; i < hi; ++i) {
          uint8_t& m = seen[(size_t)codes[i]];
          dup |= m != 0;
          m = 1;
        }
        for (int64_t i = lo; i < hi; ++i) seen[(size_t)codes[i]] = 0;
        if (dup) {
          int64_t cur = found.load();
          while (g < cur && !found.compare_exchange_weak(cur, g)) {
          }
          return;
        }
      }
    }
  });
  if (oom.load()) throw std::bad_alloc();
  const int64_t f = found.load();
  return f == INT64_MAX ? -1 : f;
}

}  // namespace ffx_host

#endif  // FFX_STRING_CODE_H_
