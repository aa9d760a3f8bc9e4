// host_io.cpp -- host-side natives of the re-ranking path: string-id tables and TREC run files.
//
// Once the arithmetic runs at HBM speed the reference's host work dominates end to end: the
// per-id Python loop of get_indices (src/fast_forward/index/util.py:29-41), the pandas string
// merges (index/base.py:296-298, :422) and read_csv / to_csv of the run
// (src/fast_forward/ranking.py:348-366, :388-409).  This file replaces those with
//   ffx_ids_*  an insertion-ordered string -> ordinal table (open addressing, FNV-1a)
//   ffx_run_*  a TREC run parsed straight into the CSR the kernels take, and written back
// No device code here; plain C++17.
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <cctype>
#include <charconv>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <numeric>
#include <string>
#include <string_view>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/ffx.h"
#include "../../include/ffx_private.h"

int ffx_fail(int code, const char* fmt, ...);  // ffx.hip: records the thread's last error

namespace {

// cores this process may really use: the affinity mask, capped by the cgroup CPU quota (a container on a
// 256-core host is often allowed 16; 128 threads on that quota crawl)
inline int usable_cores() {
  int n = (int)std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::min(n, std::max(1, CPU_COUNT(&set)));
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char quota[32];
    long period = 0;
    if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
      n = std::min(n, std::max(1, (int)((atol(quota) + period / 2) / period)));
    fclose(f);
  }
  return n;
}

// body(t) for t in 0..T-1, t = 0 on the calling thread.  If a thread cannot be started, the ones already running are
// joined before the std::system_error goes on to the caller (a std::thread destroyed while joinable ends the process).
template <class Body>
void run_threads(int T, Body&& body) {
  std::vector<std::thread> pool;
  pool.reserve(T > 1 ? (size_t)(T - 1) : 0);
  try {
    for (int t = 1; t < T; ++t) pool.emplace_back([&body, t] { body(t); });
  } catch (...) {
    for (auto& th : pool) th.join();
    throw;
  }
  body(0);
  for (auto& th : pool) th.join();
}

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

inline uint64_t fnv1a(const char* p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) {
    h ^= (unsigned char)p[i];
    h *= 1099511628211ull;
  }
  return h ^ (h >> 29);
}

}  // namespace

// insertion-ordered interning table: ordinal = order of first insertion
struct ffx_ids {
  std::vector<char> arena;
  std::vector<uint64_t> off;   // off[i]..off[i+1] = bytes of string i
  std::vector<int32_t> slots;  // open addressing, -1 = empty
  size_t mask = 0;

  ffx_ids() { off.push_back(0); rehash(1024); }
  int64_t size() const { return (int64_t)off.size() - 1; }
  std::string_view get(int32_t i) const { return {arena.data() + off[(size_t)i], (size_t)(off[(size_t)i + 1] - off[(size_t)i])}; }
  void rehash(size_t cap) {
    slots.assign(cap, -1);
    mask = cap - 1;
    for (int32_t i = 0; i < (int32_t)size(); ++i) {
      const std::string_view s = get(i);
      size_t h = fnv1a(s.data(), s.size()) & mask;
      while (slots[h] >= 0) h = (h + 1) & mask;
      slots[h] = i;
    }
  }
  int32_t find(std::string_view s) const {
    size_t h = fnv1a(s.data(), s.size()) & mask;
    while (true) {
      const int32_t i = slots[h];
      if (i < 0) return -1;
      if (get(i) == s) return i;
      h = (h + 1) & mask;
    }
  }
  int32_t intern(std::string_view s) {
    size_t h = fnv1a(s.data(), s.size()) & mask;
    while (true) {
      const int32_t i = slots[h];
      if (i < 0) break;
      if (get(i) == s) return i;
      h = (h + 1) & mask;
    }
    const int32_t id = (int32_t)size();
    arena.insert(arena.end(), s.begin(), s.end());
    off.push_back(arena.size());
    slots[h] = id;
    if ((size_t)size() * 10 > slots.size() * 6) rehash(slots.size() * 2);
    return id;
  }
};

// one parsed TREC run: rows grouped by query (order of first appearance), rows of a query in file order
struct ffx_run {
  ffx_ids qids;                     // query ids in order of first appearance
  std::vector<int64_t> pair_off;    // [Q+1]
  std::vector<int32_t> pair_ids;    // ordinal in the id table given to ffx_run_open
  std::vector<float> score;
  std::string tag;                  // last column of the first line (Ranking.name, ranking.py:409)
  const ffx_ids* ids = nullptr;     // borrowed: candidate id strings for writing
  // opened without an id table (Ranking.from_file): the candidate ids stay views into the file text
  std::string text;
  std::vector<std::string_view> pair_view;
  int32_t q_flags = 0, id_flags = 0, misc_flags = 0;
};

namespace {

// What pandas.read_csv would make of a column holding this token (it decides per column, so the
// caller ORs the classes of all tokens): an integer it prints back unchanged, another number or
// boolean (printed back differently, or only when the whole column is like that), one of its NA
// words, or plain text.
inline int token_class(std::string_view t) {
  static const char* const na_words[] = {"#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN",
                                         "<NA>", "N/A", "NA", "NULL", "NaN", "None", "n/a", "nan", "null"};
  if (t.empty()) return FFX_TOKEN_TEXT;
  // (called once per row of a run file: the common cases -- "D1234", "1234" -- are decided on their first byte)
  const char c0 = t[0];
  if (c0 == '#' || c0 == '-' || c0 == '1' || c0 == '<' || c0 == 'N' || c0 == 'n')
    for (const char* w : na_words)
      if (t == w) return FFX_TOKEN_NA;
  size_t i = 0;
  if (i < t.size() && t[i] == '-') ++i;
  const size_t digits_from = i;
  while (i < t.size() && t[i] >= '0' && t[i] <= '9') ++i;
  if (i == t.size() && i > digits_from && i - digits_from <= 18 && (t[digits_from] != '0' || (i - digits_from == 1 && digits_from == 0)))
    return FFX_TOKEN_INT;
  {
    // a number or boolean starts (after a sign) with a digit, a point, or the first letter of inf / nan / true / false
    const char c1 = (c0 == '-' || c0 == '+') ? (t.size() > 1 ? t[1] : '\0') : c0;
    const bool maybe = (c1 >= '0' && c1 <= '9') || c1 == '.' || c1 == 'i' || c1 == 'I' || c1 == 'n' || c1 == 'N' || c1 == 't' || c1 == 'T' ||
                       c1 == 'f' || c1 == 'F';
    if (!maybe) return FFX_TOKEN_TEXT;
  }
  std::string tmp(t);
  for (char& c : tmp) c = (char)tolower((unsigned char)c);
  if (tmp == "true" || tmp == "false") return FFX_TOKEN_NUMBER;
  char* endp = nullptr;
  (void)strtod(tmp.c_str(), &endp);
  if (endp == tmp.c_str() + tmp.size() && !tmp.empty()) return FFX_TOKEN_NUMBER;
  return FFX_TOKEN_TEXT;
}

// A score field as pandas.read_csv would take it: 0 = a plain decimal number or an infinity (strtod
// and pandas agree on those), 1 = one of pandas' NA words (the row is dropped, ranking.py:103),
// 2 = anything else -- hexadecimal floats, "nan(123)", "NAN", ... -- which strtod would accept but
// pandas would not read as a number: the caller rejects the file so that the pandas route decides.
inline int score_token_kind(std::string_view t) {
  size_t i = 0;
  if (i < t.size() && (t[i] == '-' || t[i] == '+')) ++i;
  const bool digit_first = i < t.size() && ((t[i] >= '0' && t[i] <= '9') || t[i] == '.');
  if (!digit_first) {  // words: pandas' NA spellings, infinities, anything else
    if (token_class(t) == FFX_TOKEN_NA) return 1;
    std::string low(t.substr(i));
    for (char& c : low) c = (char)tolower((unsigned char)c);
    return (low == "inf" || low == "infinity") ? 0 : 2;
  }
  size_t digits = 0;
  while (i < t.size() && t[i] >= '0' && t[i] <= '9') ++i, ++digits;
  if (i < t.size() && t[i] == '.') {
    ++i;
    while (i < t.size() && t[i] >= '0' && t[i] <= '9') ++i, ++digits;
  }
  if (digits == 0) return 2;
  if (i < t.size() && (t[i] == 'e' || t[i] == 'E')) {
    ++i;
    if (i < t.size() && (t[i] == '-' || t[i] == '+')) ++i;
    size_t exp_digits = 0;
    while (i < t.size() && t[i] >= '0' && t[i] <= '9') ++i, ++exp_digits;
    if (exp_digits == 0) return 2;
  }
  if (i == t.size()) return 0;
  return token_class(t) == FFX_TOKEN_NA ? 1 : 2;  // "1.#IND", "-1.#QNAN", ... start like numbers
}

inline std::string_view fixed_width(const char* base, int32_t width, int64_t i) {
  const char* p = base + (size_t)i * width;
  size_t n = 0;
  while (n < (size_t)width && p[n] != '\0') ++n;
  return {p, n};
}

// shortest round-trip decimal of a float, laid out like numpy's str(np.float32), which is
// what pandas.to_csv writes for a float32 column (ranking.py:360-366)
void append_float_repr(std::string& out, float v) {
  if (std::isnan(v)) { out += "nan"; return; }
  if (std::isinf(v)) { out += v < 0 ? "-inf" : "inf"; return; }
  char buf[48];
  auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
  std::string_view s(buf, (size_t)(r.ptr - buf));  // d[.ddd]e[+-]XX
  bool neg = false;
  if (!s.empty() && s[0] == '-') { neg = true; s.remove_prefix(1); }
  const size_t e = s.find('e');
  std::string digits;
  for (char c : s.substr(0, e)) if (c != '.') digits.push_back(c);
  int exp10 = 0;
  std::from_chars(s.data() + e + 1 + (s[e + 1] == '+' ? 1 : 0), s.data() + s.size(), exp10);
  if (neg) out.push_back('-');
  // numpy switches to scientific on the VALUE (|v| < 1e-4 or >= 1e16), not on the decimal exponent
  // of the shortest digits: float32(1e-4) = 9.99999975e-05 prints as 1e-04
  const double mag = std::fabs((double)v);
  if ((mag != 0 && mag < 1e-4) || mag >= 1e16) {
    out.push_back(digits[0]);
    if (digits.size() > 1) { out.push_back('.'); out.append(digits, 1, std::string::npos); }
    char eb[8];
    snprintf(eb, sizeof eb, "e%c%02d", exp10 < 0 ? '-' : '+', std::abs(exp10) % 1000);  // a float's decimal exponent has two digits
    out += eb;
  } else if (exp10 < 0) {
    out += "0.";
    out.append((size_t)(-exp10 - 1), '0');
    out += digits;
  } else {
    const size_t int_len = (size_t)exp10 + 1;
    if (digits.size() <= int_len) {
      out += digits;
      out.append(int_len - digits.size(), '0');
      out += ".0";
    } else {
      out.append(digits, 0, int_len);
      out.push_back('.');
      out.append(digits, int_len, std::string::npos);
    }
  }
}

// The run writers format lines by threads and write them in order: queries are handed out in batches of about
// 64K lines per thread; while one round of buffers is written to the file the next round is being formatted.
// format(q_lo, q_hi, out) appends the lines of queries q_lo .. q_hi-1 to `out`.
template <class Format>
int write_formatted(FILE* f, const char* path, int64_t n_queries, int64_t n_rows, Format format) {
  const int T = (int)std::min<int64_t>(std::max(1, std::min(32, usable_cores())), std::max<int64_t>(1, n_rows >> 15));
  std::vector<std::string> bufs[2] = {std::vector<std::string>((size_t)T), std::vector<std::string>((size_t)T)};
  int used[2] = {0, 0};
  bool ok = true;
  std::atomic<bool> oom(false);
  try {
    const int64_t avg = std::max<int64_t>(1, n_rows / std::max<int64_t>(1, n_queries));
    const int64_t q_per_thread = std::max<int64_t>(1, (64 << 10) / avg);
    int round = 0;
    for (int64_t q0 = 0; (q0 < n_queries || used[(round + 1) & 1] > 0) && ok; q0 += q_per_thread * T, ++round) {
      std::vector<std::string>& mine = bufs[round & 1];
      std::vector<std::thread> pool;
      pool.reserve((size_t)T);
      used[round & 1] = 0;
      try {
        for (int t = 0; t < T; ++t) {
          const int64_t lo = q0 + t * q_per_thread, hi = std::min(n_queries, lo + q_per_thread);
          if (lo >= hi) break;
          pool.emplace_back([&format, &oom, lo, hi, buf = &mine[(size_t)t]] {
            try {
              format(lo, hi, *buf);
            } catch (const std::bad_alloc&) {  // (an exception leaving a thread would end the process)
              oom.store(true);
            }
          });
          ++used[round & 1];
        }
      } catch (...) {  // join what was started before the error goes on
        for (auto& th : pool) th.join();
        throw;
      }
      // meanwhile: the previous round's buffers go to the file
      std::vector<std::string>& prev = bufs[(round + 1) & 1];
      for (int t = 0; t < used[(round + 1) & 1] && ok; ++t)
        ok = prev[(size_t)t].empty() || fwrite(prev[(size_t)t].data(), 1, prev[(size_t)t].size(), f) == prev[(size_t)t].size();
      used[(round + 1) & 1] = 0;
      for (auto& th : pool) th.join();
      if (oom.load()) throw std::bad_alloc();
    }
  } catch (const std::bad_alloc&) {
    fclose(f);
    return ffx_fail(FFX_OOM, "host allocation failed while writing %s", path);
  } catch (const std::system_error&) {
    fclose(f);
    return ffx_fail(FFX_OOM, "could not start host threads while writing %s", path);
  }
  if (!ok) {
    fclose(f);
    return ffx_fail(FFX_BAD_ARG, "write to %s failed", path);
  }
  return fclose(f) == 0 ? FFX_OK : ffx_fail(FFX_BAD_ARG, "write to %s failed", path);
}

}  // namespace

// steps 2 - 5 of ffx_strings_code for n strings given as view(i) -> std::string_view (hashes, shards, per-shard
// numbering, global numbers + the distinct strings joined by sep)
template <class View>
static int strings_code_views(int64_t n, int T, View&& view, char sep, int32_t* codes, char** names_blob, int64_t* names_bytes, int64_t* n_names) {
  constexpr int S = 256;
  auto in_threads = [&](auto&& body) { run_threads(T, body); };
  {
    // 2. hashes, and how many rows of every row range fall into every shard
    std::vector<uint64_t> hash((size_t)n);
    std::vector<int64_t> count((size_t)T * S, 0);
    auto row_range = [&](int t) { return std::pair<int64_t, int64_t>(n * t / T, n * (t + 1) / T); };
    in_threads([&](int t) {
      const auto [lo, hi] = row_range(t);
      int64_t* c = count.data() + (size_t)t * S;
      for (int64_t i = lo; i < hi; ++i) {
        const std::string_view v = view(i);
        uint64_t h = fnv1a(v.data(), v.size());
        h ^= h >> 29;
        h *= 0xbf58476d1ce4e5b9ull;  // fnv's high bits are weak on short numeric strings: mix before taking the shard from them
        h ^= h >> 32;
        hash[(size_t)i] = h;
        ++c[h >> 56];
      }
    });
    // 3. rows dealt into shards, row order kept inside a shard
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
    std::vector<int32_t> rows((size_t)n);
    in_threads([&](int t) {
      const auto [lo, hi] = row_range(t);
      int64_t* at = count.data() + (size_t)t * S;
      for (int64_t i = lo; i < hi; ++i) rows[(size_t)at[hash[(size_t)i] >> 56]++] = (int32_t)i;
    });
    // 4. every shard numbered on its own; codes[] holds shard-local numbers for now
    std::vector<std::vector<int32_t>> firsts((size_t)S);  // first row of every distinct string of the shard, in number order
    std::vector<int64_t> shard_bytes((size_t)S, 0);
    std::atomic<int> next_shard(0);
    std::atomic<bool> oom(false);
    in_threads([&](int) {
      try {
        std::vector<int32_t> slot;
        for (;;) {
          const int s = next_shard.fetch_add(1);
          if (s >= S) return;
          const int64_t lo = shard_off[(size_t)s], hi = shard_off[(size_t)s + 1];
          size_t cap = 16;
          while (cap < (size_t)(hi - lo) * 2) cap <<= 1;
          slot.assign(cap, -1);  // slot -> shard-local number
          auto& first = firsts[(size_t)s];
          int64_t bytes = 0;
          for (int64_t k = lo; k < hi; ++k) {
            const int32_t i = rows[(size_t)k];
            const uint64_t h = hash[(size_t)i];
            const std::string_view v = view(i);
            size_t at = (size_t)(h >> 8) & (cap - 1);
            for (;;) {
              const int32_t have = slot[at];
              if (have < 0) {
                slot[at] = (int32_t)first.size();
                codes[i] = (int32_t)first.size();
                first.push_back(i);
                bytes += (int64_t)v.size();
                break;
              }
              const int32_t j = first[(size_t)have];
              if (hash[(size_t)j] == h && view(j) == v) {
                codes[i] = have;
                break;
              }
              at = (at + 1) & (cap - 1);
            }
          }
          shard_bytes[(size_t)s] = bytes;
        }
      } catch (const std::bad_alloc&) {
        oom.store(true);
      }
    });
    if (oom.load()) return ffx_fail(FFX_OOM, "host allocation failed");
    // 5. shard-local numbers -> global numbers; the names written shard by shard
    std::vector<int64_t> base((size_t)S + 1, 0), at_byte((size_t)S + 1, 0);
    for (int s = 0; s < S; ++s) {
      base[(size_t)s + 1] = base[(size_t)s] + (int64_t)firsts[(size_t)s].size();
      at_byte[(size_t)s + 1] = at_byte[(size_t)s] + shard_bytes[(size_t)s] + (int64_t)firsts[(size_t)s].size();  // + one separator each
    }
    const int64_t total = at_byte[(size_t)S];  // one separator too many (after the last name)
    char* blob = static_cast<char*>(malloc((size_t)std::max<int64_t>(total, 1)));
    if (!blob) return ffx_fail(FFX_OOM, "host allocation failed");
    next_shard.store(0);
    in_threads([&](int) {
      try {
      for (;;) {
        const int s = next_shard.fetch_add(1);
        if (s >= S) return;
        const int32_t b = (int32_t)base[(size_t)s];
        // Inside the shard the strings are numbered by (hash, bytes), not by first appearance: the numbering then
        // depends on the SET of strings only -- not on the order of the rows, the thread count or (for strings that
        // were gathered through their object addresses first) the process.  Ranks of a sharded index number the same
        // ids alike and can talk about them by number.
        auto& first = firsts[(size_t)s];
        std::vector<int32_t> by_content(first.size()), renumber(first.size());
        for (size_t j = 0; j < first.size(); ++j) by_content[j] = (int32_t)j;
        std::sort(by_content.begin(), by_content.end(), [&](int32_t x, int32_t y) {
          const uint64_t hx = hash[(size_t)first[(size_t)x]], hy = hash[(size_t)first[(size_t)y]];
          return hx != hy ? hx < hy : view(first[(size_t)x]) < view(first[(size_t)y]);
        });
        for (size_t j = 0; j < first.size(); ++j) renumber[(size_t)by_content[j]] = (int32_t)j;
        {
          std::vector<int32_t> sorted(first.size());
          for (size_t j = 0; j < first.size(); ++j) sorted[j] = first[(size_t)by_content[j]];
          first.swap(sorted);
        }
        for (int64_t k = shard_off[(size_t)s]; k < shard_off[(size_t)s + 1]; ++k) {
          int32_t& c = codes[rows[(size_t)k]];
          c = renumber[(size_t)c] + b;
        }
        char* w = blob + at_byte[(size_t)s];
        for (const int32_t i : firsts[(size_t)s]) {
          const std::string_view v = view(i);
          memcpy(w, v.data(), v.size());
          w += v.size();
          *w++ = sep;
        }
      }
      } catch (const std::bad_alloc&) {  // (an exception leaving a thread would end the process)
        oom.store(true);
      }
    });
    if (oom.load()) {
      free(blob);
      return ffx_fail(FFX_OOM, "host allocation failed");
    }
    *names_blob = blob;
    *names_bytes = total - 1;
    *n_names = base[(size_t)S];
    return FFX_OK;
  }
}

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
    size_t h[B];
    int32_t cand[B];
    for (int64_t done = 0; done < owned;) {
      const int nb = (int)std::min<int64_t>(B, owned - done);
      for (int b2 = 0; b2 < nb; ++b2) {
        const char* q = static_cast<const char*>(memchr(p, sep, (size_t)(buf_end - p)));
        const char* e = q ? q : buf_end;
        sv[b2] = std::string_view(p, (size_t)(e - p));
        h[b2] = fnv1a(p, (size_t)(e - p)) & t->mask;
        __builtin_prefetch(&t->slots[h[b2]]);
        p = e + 1;
      }
      for (int b2 = 0; b2 < nb; ++b2) {
        cand[b2] = t->slots[h[b2]];
        if (cand[b2] >= 0) __builtin_prefetch(&t->off[(size_t)cand[b2]]);
      }
      for (int b2 = 0; b2 < nb; ++b2)
        if (cand[b2] >= 0) __builtin_prefetch(t->arena.data() + t->off[(size_t)cand[b2]]);
      for (int b2 = 0; b2 < nb; ++b2) {
        size_t hh = h[b2];
        int32_t o = -1;
        while (true) {
          const int32_t c = t->slots[hh];
          if (c < 0) break;
          if (t->get(c) == sv[b2]) {
            o = c;
            break;
          }
          hh = (hh + 1) & t->mask;
        }
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
    return strings_code_views(n, T, view, sep, codes, names_blob, names_bytes, n_names);
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
    std::vector<int32_t> rows((size_t)n);
    run_threads(T, [&](int t) {
      const auto [lo, hi] = row_range(t);
      int64_t* at = count.data() + (size_t)t * S;
      for (int64_t i = lo; i < hi; ++i) rows[(size_t)at[shard[(size_t)i]]++] = (int32_t)i;
    });
    std::vector<std::vector<int32_t>> firsts((size_t)S);
    std::atomic<int> next_shard(0);
    std::atomic<bool> oom(false);
    run_threads(T, [&](int) {
      try {
        std::vector<int32_t> slot, by_value, renumber;
        for (;;) {
          const int s = next_shard.fetch_add(1);
          if (s >= S) return;
          const int64_t lo = shard_off[(size_t)s], hi = shard_off[(size_t)s + 1];
          size_t cap = 16;
          while (cap < (size_t)(hi - lo) * 2) cap <<= 1;
          slot.assign(cap, -1);
          auto& first = firsts[(size_t)s];
          for (int64_t k = lo; k < hi; ++k) {
            const int32_t i = rows[(size_t)k];
            const uint64_t v = values[i];
            size_t at = (size_t)(mix(v) >> 8) & (cap - 1);
            for (;;) {
              const int32_t have = slot[at];
              if (have < 0) {
                slot[at] = (int32_t)first.size();
                codes[i] = (int32_t)first.size();
                first.push_back(i);
                break;
              }
              if (values[first[(size_t)have]] == v) {
                codes[i] = have;
                break;
              }
              at = (at + 1) & (cap - 1);
            }
          }
          // the shard's distinct values in ascending order
          by_value.resize(first.size());
          renumber.resize(first.size());
          for (size_t j = 0; j < first.size(); ++j) by_value[j] = (int32_t)j;
          std::sort(by_value.begin(), by_value.end(), [&](int32_t x, int32_t y) { return values[first[(size_t)x]] < values[first[(size_t)y]]; });
          for (size_t j = 0; j < first.size(); ++j) renumber[(size_t)by_value[j]] = (int32_t)j;
          std::vector<int32_t> sorted(first.size());
          for (size_t j = 0; j < first.size(); ++j) sorted[j] = first[(size_t)by_value[j]];
          first.swap(sorted);
          for (int64_t k = lo; k < hi; ++k) {
            int32_t& c = codes[rows[(size_t)k]];
            c = renumber[(size_t)c];
          }
        }
      } catch (const std::bad_alloc&) {
        oom.store(true);
      }
    });
    if (oom.load()) return ffx_fail(FFX_OOM, "host allocation failed");
    std::vector<int64_t> base((size_t)S + 1, 0);
    for (int s = 0; s < S; ++s) base[(size_t)s + 1] = base[(size_t)s] + (int64_t)firsts[(size_t)s].size();
    next_shard.store(0);
    run_threads(T, [&](int) {
      for (;;) {
        const int s = next_shard.fetch_add(1);
        if (s >= S) return;
        const int64_t b = base[(size_t)s];
        if (b != 0)
          for (int64_t k = shard_off[(size_t)s]; k < shard_off[(size_t)s + 1]; ++k) codes[rows[(size_t)k]] += (int32_t)b;
        int64_t w = b;
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
  if (!pair_off || n_queries < 0 || pair_off[0] != 0) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (dst_off)
    for (int64_t q = 0; q < n_queries; ++q)
      if (dst_off[q] < 0 || dst_off[q] + (pair_off[q + 1] - pair_off[q]) > pair_off[n_queries]) return ffx_fail(FFX_BAD_ARG, "destination of query %lld lies outside the output", (long long)q);
  for (int64_t q = 0; q < n_queries; ++q)
    if (pair_off[q + 1] < pair_off[q]) return ffx_fail(FFX_BAD_ARG, "pair_off not monotone at query %lld", (long long)q);
  const int64_t P = pair_off[n_queries];
  if (P > INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "too many pairs");
  if (P > 0 && (!scores || !out_order)) return ffx_fail(FFX_BAD_ARG, "NULL argument");
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
          int32_t* dst = out_order + (dst_off ? dst_off[q] : lo);
          for (int64_t i = 0; i < n; ++i) dst[i] = (int32_t)(lo + (int64_t)(keys[(size_t)i] & 0xFFFFFFFFu));
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

// ---- TREC run files ---------------------------------------------------------------------------------
// Line format (Ranking.from_file, ranking.py:400-409): q_id Q0 id rank score name, any whitespace.
// Candidate ids are resolved against `ids` while parsing; the first unknown id is FFX_NOT_FOUND with
// its 1-based line number in *bad_line (the reference's IndexError, index/util.py:38-39).
// A (query, id) pair that appears twice is FFX_BAD_ARG (ValueError, ranking.py:95-98).
int ffx_run_open(const char* path, const ffx_ids* ids, ffx_run** out, int64_t* bad_line) {
  if (bad_line) *bad_line = -1;
  if (!path || !out) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  *out = nullptr;
  const bool standalone = ids == nullptr;  // keep the id strings instead of resolving them
  FILE* f = fopen(path, "rb");
  if (!f) return ffx_fail(FFX_BAD_ARG, "cannot open %s", path);
  std::string data;
  try {
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    data.resize(sz > 0 ? (size_t)sz : 0);
    if (sz > 0 && fread(data.data(), 1, (size_t)sz, f) != (size_t)sz) {
      fclose(f);
      return ffx_fail(FFX_BAD_ARG, "short read on %s", path);
    }
  } catch (const std::bad_alloc&) {
    fclose(f);
    return ffx_fail(FFX_OOM, "run file does not fit in memory");
  }
  fclose(f);

  ffx_run* run = new (std::nothrow) ffx_run();
  if (!run) return ffx_fail(FFX_OOM, "host allocation failed");
  run->ids = ids;
  std::vector<int32_t> row_q, row_id;
  std::vector<std::string_view> row_view;
  std::vector<float> row_s;
  int rc = FFX_OK;
  // The file is cut into pieces at line ends and the pieces are tokenised, resolved against the
  // (read-only) id table and converted in parallel; query ids are run-length encoded per piece
  // and interned afterwards in file order, so numbering and errors are those of a serial pass.
  struct Piece {
    const char *begin = nullptr, *end = nullptr;
    std::vector<int32_t> id;
    std::vector<std::string_view> id_view;  // standalone mode
    int32_t q_flags = 0, id_flags = 0, misc_flags = 0;
    std::vector<float> score;
    std::vector<std::pair<std::string_view, int64_t>> q_runs;  // (query id, consecutive rows)
    std::string_view tag;
    int64_t lines = 0;
    int err = FFX_OK;  // first problem of the piece
    int64_t err_line = 0;
    std::string err_text;
    bool oom = false;
  };
  if (standalone) {
    run->text = std::move(data);  // the id views point into it
    if (memchr(run->text.data(), '"', run->text.size())) run->misc_flags |= FFX_RUN_QUOTES;
  }
  const std::string& text = standalone ? run->text : data;
  const char* const base = text.data();
  const char* const end = base + text.size();
  const int T = (int)std::min<size_t>((size_t)std::max(1, std::min(32, usable_cores())), std::max<size_t>(1, text.size() >> 20));
  std::vector<Piece> pieces((size_t)T);
  {
    const char* cut = base;
    for (int t = 0; t < T; ++t) {
      pieces[(size_t)t].begin = cut;
      const char* want = t + 1 == T ? end : base + text.size() * (size_t)(t + 1) / (size_t)T;
      if (want < cut) want = cut;
      if (want < end) {
        const char* nl = (const char*)memchr(want, '\n', (size_t)(end - want));
        want = nl ? nl + 1 : end;
      }
      pieces[(size_t)t].end = cut = want;
    }
  }
  auto parse = [&](Piece& pc) {
    auto is_space = [](char c) { return c == ' ' || c == '\t' || c == '\r'; };
    try {
      const char* p = pc.begin;
      while (p < pc.end) {
        const char* eol = (const char*)memchr(p, '\n', (size_t)(pc.end - p));
        if (!eol) eol = pc.end;
        ++pc.lines;
        std::string_view field[6];
        int nf = 0;
        const char* c = p;
        while (c < eol && nf < 6) {
          while (c < eol && is_space(*c)) ++c;
          if (c >= eol) break;
          const char* s0 = c;
          while (c < eol && !is_space(*c)) ++c;
          field[nf++] = {s0, (size_t)(c - s0)};
        }
        p = eol + 1;
        if (nf == 0) continue;  // blank line
        if (nf < 5) {
          pc.err = FFX_BAD_ARG, pc.err_line = pc.lines, pc.err_text = "expected 'q_id Q0 id rank score [name]'";
          return;
        }
        const int score_kind = score_token_kind(field[4]);
        if (score_kind == 2) {
          pc.err = FFX_BAD_ARG, pc.err_line = pc.lines, pc.err_text = "bad score '" + std::string(field[4]) + "'";
          return;
        }
        if (score_kind == 1) continue;  // a missing score: the reference drops the row (ranking.py:103)
        int32_t o = 0;
        if (standalone) {
          pc.id_flags |= token_class(field[2]);
          if (pc.q_runs.empty() || pc.q_runs.back().first != field[0]) pc.q_flags |= token_class(field[0]);
          if (nf < 6) pc.misc_flags |= FFX_RUN_SHORT_LINE;
          for (const char* x = c; x < eol; ++x)
            if (!is_space(*x)) pc.misc_flags |= FFX_RUN_LONG_LINE;
        } else {
          o = ids->find(field[2]);
          if (o < 0) {
            pc.err = FFX_NOT_FOUND, pc.err_line = pc.lines, pc.err_text.assign(field[2]);
            return;
          }
        }
        // correctly rounded double, then float32 (what read_csv + the constructor's cast give); from_chars is
        // several times faster than strtod and needs no terminator; it takes no leading '+'
        double sc = 0.0;
        {
          const char* s0 = field[4].data();
          const char* s1 = s0 + field[4].size();
          if (s0 < s1 && *s0 == '+') ++s0;
          const auto res = std::from_chars(s0, s1, sc);
          if (res.ec == std::errc::result_out_of_range) {  // strtod's answer for these: +-inf / +-0
            char* endp = nullptr;
            sc = strtod(field[4].data(), &endp);  // (the buffer ends in NUL and a field ends at white space)
            if (endp != s1) {
              pc.err = FFX_BAD_ARG, pc.err_line = pc.lines, pc.err_text = "bad score '" + std::string(field[4]) + "'";
              return;
            }
          } else if (res.ec != std::errc() || res.ptr != s1) {
            pc.err = FFX_BAD_ARG, pc.err_line = pc.lines, pc.err_text = "bad score '" + std::string(field[4]) + "'";
            return;
          }
        }
        if (pc.tag.data() == nullptr && nf >= 6) pc.tag = field[5];
        if (pc.q_runs.empty() || pc.q_runs.back().first != field[0]) pc.q_runs.emplace_back(field[0], 0);
        ++pc.q_runs.back().second;
        if (standalone)
          pc.id_view.push_back(field[2]);
        else
          pc.id.push_back(o);
        pc.score.push_back((float)sc);  // scores become float32 (Ranking dtype, ranking.py:72)
      }
    } catch (const std::bad_alloc&) {
      pc.oom = true;
    }
  };
  try {
    run_threads(T, [&](int t) { parse(pieces[(size_t)t]); });
    int64_t lines_before = 0;
    for (Piece& pc : pieces) {
      if (pc.oom) throw std::bad_alloc();
      if (pc.err != FFX_OK) {
        const int64_t line_no = lines_before + pc.err_line;
        if (bad_line) *bad_line = line_no;
        rc = pc.err == FFX_NOT_FOUND ? ffx_fail(FFX_NOT_FOUND, "ID %s not found in the index.", pc.err_text.c_str())
                                     : ffx_fail(FFX_BAD_ARG, "%s:%lld: %s", path, (long long)line_no, pc.err_text.c_str());
        break;
      }
      lines_before += pc.lines;
      if (run->tag.empty() && pc.tag.data() != nullptr) run->tag.assign(pc.tag);
      row_id.insert(row_id.end(), pc.id.begin(), pc.id.end());
      row_view.insert(row_view.end(), pc.id_view.begin(), pc.id_view.end());
      run->q_flags |= pc.q_flags;
      run->id_flags |= pc.id_flags;
      run->misc_flags |= pc.misc_flags;
      row_s.insert(row_s.end(), pc.score.begin(), pc.score.end());
      for (const auto& qr : pc.q_runs) row_q.insert(row_q.end(), (size_t)qr.second, run->qids.intern(qr.first));
    }
    if (rc == FFX_OK) {
      const int64_t nq = run->qids.size(), n = (int64_t)row_q.size();
      run->pair_off.assign((size_t)nq + 1, 0);
      for (int32_t q : row_q) ++run->pair_off[(size_t)q + 1];
      for (int64_t q = 0; q < nq; ++q) run->pair_off[(size_t)q + 1] += run->pair_off[(size_t)q];
      run->pair_ids.resize(standalone ? 0 : (size_t)n);
      run->pair_view.resize(standalone ? (size_t)n : 0);
      run->score.resize((size_t)n);
      std::vector<int64_t> cursor(run->pair_off.begin(), run->pair_off.end() - 1);
      for (int64_t i = 0; i < n; ++i) {  // stable: rows of a query keep file order
        const int64_t dst = cursor[(size_t)row_q[(size_t)i]]++;
        if (standalone)
          run->pair_view[(size_t)dst] = row_view[(size_t)i];
        else
          run->pair_ids[(size_t)dst] = row_id[(size_t)i];
        run->score[(size_t)dst] = row_s[(size_t)i];
      }
      // duplicate (query, id) pairs: one mark per ordinal, set and cleared query by query
      std::atomic<bool> dup{false}, scan_oom{false};
      auto scan = [&](int64_t q_lo, int64_t q_hi) {
        if (standalone) {  // no ordinals: order each query's id strings and look at neighbours
          std::vector<std::string_view> tmp;
          try {
            for (int64_t q = q_lo; q < q_hi && !dup.load(std::memory_order_relaxed); ++q) {
              tmp.assign(run->pair_view.begin() + run->pair_off[(size_t)q], run->pair_view.begin() + run->pair_off[(size_t)q + 1]);
              std::sort(tmp.begin(), tmp.end());
              if (std::adjacent_find(tmp.begin(), tmp.end()) != tmp.end()) dup.store(true);
            }
          } catch (const std::bad_alloc&) {
            scan_oom.store(true);
          }
          return;
        }
        std::vector<uint8_t> seen;
        try {
          seen.assign((size_t)ids->size(), 0);
        } catch (const std::bad_alloc&) {
          scan_oom.store(true);
          return;
        }
        for (int64_t q = q_lo; q < q_hi && !dup.load(std::memory_order_relaxed); ++q) {
          const int64_t lo = run->pair_off[(size_t)q], hi = run->pair_off[(size_t)q + 1];
          for (int64_t i = lo; i < hi; ++i) {
            uint8_t& m = seen[(size_t)run->pair_ids[(size_t)i]];
            if (m) dup.store(true);
            m = 1;
          }
          for (int64_t i = lo; i < hi; ++i) seen[(size_t)run->pair_ids[(size_t)i]] = 0;
        }
      };
      const int TD = (int)std::min<int64_t>(T, std::max<int64_t>(1, n >> 16));
      run_threads(TD, [&](int t) { scan(nq * t / TD, nq * (t + 1) / TD); });
      if (scan_oom.load()) throw std::bad_alloc();
      if (dup.load()) rc = ffx_fail(FFX_BAD_ARG, "Only one score per query-document/passage pair is allowed.");
    }
  } catch (const std::bad_alloc&) {
    rc = ffx_fail(FFX_OOM, "host allocation failed while parsing %s", path);
  } catch (const std::system_error&) {
    rc = ffx_fail(FFX_OOM, "could not start host threads while parsing %s", path);
  }
  if (rc != FFX_OK) {
    delete run;
    return rc;
  }
  *out = run;
  return FFX_OK;
}

int ffx_run_close(ffx_run* run) {
  delete run;
  return FFX_OK;
}

int ffx_run_counts(const ffx_run* run, int64_t* n_queries, int64_t* n_pairs) {
  if (!run) return ffx_fail(FFX_BAD_ARG, "NULL run");
  if (n_queries) *n_queries = run->qids.size();
  if (n_pairs) *n_pairs = (int64_t)run->score.size();
  return FFX_OK;
}

int ffx_run_query_id(const ffx_run* run, int64_t q, const char** str, int32_t* len) {
  if (!run) return ffx_fail(FFX_BAD_ARG, "NULL run");
  return ffx_ids_get(&run->qids, (int32_t)q, str, len);
}

int ffx_run_tag(const ffx_run* run, const char** str, int32_t* len) {
  if (!run || !str || !len) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  *str = run->tag.data();
  *len = (int32_t)run->tag.size();
  return FFX_OK;
}

// copy out the CSR; with cutoff > 0 only each query's `cutoff` best-scoring candidates are kept
// (Ranking.cut, ranking.py:279-291: order by score descending, ties in file order)
int ffx_run_export(ffx_run* run, int64_t cutoff, int64_t* pair_off, int32_t* pair_ids, float* sparse, int64_t* n_pairs_out) {
  if (!run || !pair_off) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  const int64_t nq = run->qids.size();
  if (!run->ids && (cutoff > 0 || pair_ids)) return ffx_fail(FFX_BAD_ARG, "a run opened without an id table has no ordinals to export or cut");
  if (cutoff > 0) {  // trim in place, once
    std::vector<int64_t> new_off((size_t)nq + 1, 0);
    std::vector<int32_t> ids2;
    std::vector<float> sc2;
    std::vector<int64_t> idx;
    for (int64_t q = 0; q < nq; ++q) {
      const int64_t lo = run->pair_off[(size_t)q], hi = run->pair_off[(size_t)q + 1];
      idx.resize((size_t)(hi - lo));
      std::iota(idx.begin(), idx.end(), lo);
      if ((int64_t)idx.size() > cutoff) {
        std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return run->score[(size_t)a] > run->score[(size_t)b]; });
        idx.resize((size_t)cutoff);
        std::sort(idx.begin(), idx.end());  // survivors back in file order
      }
      for (int64_t i : idx) {
        ids2.push_back(run->pair_ids[(size_t)i]);
        sc2.push_back(run->score[(size_t)i]);
      }
      new_off[(size_t)q + 1] = (int64_t)ids2.size();
    }
    run->pair_off.swap(new_off);
    run->pair_ids.swap(ids2);
    run->score.swap(sc2);
  }
  const int64_t n = (int64_t)run->score.size();
  if (n_pairs_out) *n_pairs_out = n;
  memcpy(pair_off, run->pair_off.data(), (size_t)(nq + 1) * sizeof(int64_t));
  if (pair_ids && n > 0) memcpy(pair_ids, run->pair_ids.data(), (size_t)n * sizeof(int32_t));  // (an empty vector's data() may be NULL)
  if (sparse && n > 0) memcpy(sparse, run->score.data(), (size_t)n * sizeof(float));
  return FFX_OK;
}

// write `q_id Q0 id rank score tag` (tab separated; Ranking.save, ranking.py:348-366): queries in
// descending q_id order, candidates of a query in the order given by `order` (positions into the
// CSR, e.g. the device sort of ffx_plan_run); keep > 0 writes only the first `keep` of each query.
// standalone runs (opened without an id table): what pandas would make of the q_id / id columns
// (FFX_TOKEN_* classes OR-ed over the column) and of the file's shape (FFX_RUN_* bits)
int ffx_run_flags(const ffx_run* run, int32_t* q_flags, int32_t* id_flags, int32_t* misc_flags) {
  if (!run) return ffx_fail(FFX_BAD_ARG, "NULL run");
  if (q_flags) *q_flags = run->q_flags;
  if (id_flags) *id_flags = run->id_flags;
  if (misc_flags) *misc_flags = run->misc_flags;
  return FFX_OK;
}

// standalone runs: the candidate ids in CSR order (grouped by query, file order inside), joined by
// '\n'.  Call with buf == NULL to learn the size.
int ffx_run_ids_blob(const ffx_run* run, char* buf, int64_t cap, int64_t* needed) {
  if (!run || !needed) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  if (run->ids) return ffx_fail(FFX_BAD_ARG, "this run was resolved against an id table: it holds ordinals, not strings");
  int64_t total = 0;
  for (const std::string_view& v : run->pair_view) total += (int64_t)v.size() + 1;
  if (total > 0) --total;  // no separator after the last one
  *needed = total;
  if (!buf) return FFX_OK;
  if (cap < total) return ffx_fail(FFX_BAD_ARG, "buffer too small");
  char* w = buf;
  for (size_t i = 0; i < run->pair_view.size(); ++i) {
    if (i) *w++ = '\n';
    memcpy(w, run->pair_view[i].data(), run->pair_view[i].size());
    w += run->pair_view[i].size();
  }
  return FFX_OK;
}

int ffx_run_write(const ffx_run* run, const char* path, const float* scores, const int32_t* order, int64_t keep,
                  const char* tag) {
  if (!run || !path || !scores || !order) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  if (!run->ids) return ffx_fail(FFX_BAD_ARG, "this run was opened without an id table");
  FILE* f = fopen(path, "wb");
  if (!f) return ffx_fail(FFX_BAD_ARG, "cannot create %s", path);
  const int64_t nq = run->qids.size();
  std::vector<int64_t> qorder((size_t)nq);
  std::iota(qorder.begin(), qorder.end(), 0);
  std::sort(qorder.begin(), qorder.end(), [&](int64_t a, int64_t b) { return run->qids.get((int32_t)a) > run->qids.get((int32_t)b); });
  const std::string name = tag ? tag : run->tag;
  // formatting is the cost here (a shortest-round-trip float per line): queries are handed to a
  // few threads in batches, each fills its own buffer, and the buffers are written in order
  auto format = [&](int64_t q_lo, int64_t q_hi, std::string& out) {
    out.clear();
    for (int64_t qi = q_lo; qi < q_hi; ++qi) {
      const int64_t q = qorder[(size_t)qi];
      const std::string_view qid = run->qids.get((int32_t)q);
      const int64_t lo = run->pair_off[(size_t)q], hi = run->pair_off[(size_t)q + 1];
      const int64_t n = (keep > 0 && hi - lo > keep) ? keep : hi - lo;
      for (int64_t r = 0; r < n; ++r) {
        const int64_t p = order[lo + r];
        const std::string_view id = run->ids->get(run->pair_ids[(size_t)p]);
        out.append(qid).append("\tQ0\t").append(id).push_back('\t');
        out += std::to_string(r + 1);
        out.push_back('\t');
        append_float_repr(out, scores[p]);
        out.push_back('\t');
        out += name;
        out.push_back('\n');
      }
    }
  };
  const int64_t n_rows = run->pair_off.empty() ? 0 : run->pair_off.back();
  return write_formatted(f, path, nq, n_rows, format);
}

// Ranking.save (ranking.py:348-366) for a frame that is already grouped by query: the q_id and id
// columns arrive as two '\n'-joined buffers (n_queries names; n_pairs ids in frame order), pair_off
// gives every query's rows.  Lines "q_id\tQ0\tid\trank\tscore\tname", rank = 1-based position inside
// the query, score formatted like pandas' to_csv of a float32 column.  Formatted by threads.
}  // extern "C" (the writers below share helpers in an unnamed namespace)

namespace {
// n strings out of a buffer where the byte '\n' stands between them
bool split_lines(const char* blob, int64_t bytes, int64_t n, std::vector<std::string_view>& out) {
  out.clear();
  if (n == 0) return bytes == 0;
  out.reserve((size_t)n);
  const char* p = blob;
  const char* end = blob + bytes;
  while ((int64_t)out.size() < n) {
    const char* q = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
    if (!q) {
      out.emplace_back(p, (size_t)(end - p));
      p = end + 1;
      break;
    }
    out.emplace_back(p, (size_t)(q - p));
    p = q + 1;
  }
  return (int64_t)out.size() == n && p == end + 1;
}

// Ranking.save (ranking.py:348-366): rows grouped by query, rank = position inside the query; formatted by
// threads, written in order.  id_at(p) is the candidate id of row p.
template <class IdAt>
int save_rows(const char* path, const std::vector<std::string_view>& q_names, int64_t n_queries, const int64_t* pair_off, int64_t n_pairs,
              IdAt id_at, const float* scores, const char* tag) {
  FILE* f = fopen(path, "wb");
  if (!f) return ffx_fail(FFX_BAD_ARG, "cannot create %s", path);
  const std::string name = tag;
  auto format = [&](int64_t q_lo, int64_t q_hi, std::string& out) {
    out.clear();
    for (int64_t q = q_lo; q < q_hi; ++q) {
      for (int64_t p = pair_off[q]; p < pair_off[q + 1]; ++p) {
        out.append(q_names[(size_t)q]).append("\tQ0\t").append(id_at(p)).push_back('\t');
        out += std::to_string(p - pair_off[q] + 1);
        out.push_back('\t');
        append_float_repr(out, scores[p]);
        out.push_back('\t');
        out += name;
        out.push_back('\n');
      }
    }
  };
  return write_formatted(f, path, n_queries, n_pairs, format);
}

int check_save_args(const char* path, const int64_t* pair_off, int64_t n_queries, int64_t n_pairs, const char* tag) {
  if (!path || !pair_off || n_queries < 0 || n_pairs < 0 || !tag) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (pair_off[0] != 0 || pair_off[n_queries] != n_pairs) return ffx_fail(FFX_BAD_ARG, "pair_off does not cover the rows");
  for (int64_t q = 0; q < n_queries; ++q)
    if (pair_off[q + 1] < pair_off[q]) return ffx_fail(FFX_BAD_ARG, "pair_off not monotone at query %lld", (long long)q);
  return FFX_OK;
}
}  // namespace

extern "C" {

int ffx_run_save(const char* path, const char* q_blob, int64_t q_bytes, int64_t n_queries, const int64_t* pair_off,
                 const char* id_blob, int64_t id_bytes, int64_t n_pairs, const float* scores, const char* tag) {
  if (int rc = check_save_args(path, pair_off, n_queries, n_pairs, tag)) return rc;
  if (n_pairs > 0 && (!id_blob || !scores || !q_blob)) return ffx_fail(FFX_BAD_ARG, "bad argument");
  std::vector<std::string_view> q_names, ids;
  try {
    if (!split_lines(q_blob, q_bytes, n_queries, q_names) || !split_lines(id_blob, id_bytes, n_pairs, ids))
      return ffx_fail(FFX_BAD_ARG, "a buffer does not hold the announced number of strings");
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  }
  return save_rows(path, q_names, n_queries, pair_off, n_pairs, [&](int64_t p) { return ids[(size_t)p]; }, scores, tag);
}

// The same for rows whose candidate ids are CODES into a table of names (a run as the device hands it back:
// Ranking._assemble_coded): row p is written with name_blob's string number id_codes[p].
int ffx_run_save_coded(const char* path, const char* q_blob, int64_t q_bytes, int64_t n_queries, const int64_t* pair_off,
                       const char* name_blob, int64_t name_bytes, int64_t n_names, const int32_t* id_codes, int64_t n_pairs,
                       const float* scores, const char* tag) {
  if (int rc = check_save_args(path, pair_off, n_queries, n_pairs, tag)) return rc;
  if (n_names < 0 || (n_pairs > 0 && (!name_blob || !id_codes || !scores || !q_blob))) return ffx_fail(FFX_BAD_ARG, "bad argument");
  std::vector<std::string_view> q_names, names;
  try {
    if (!split_lines(q_blob, q_bytes, n_queries, q_names) || !split_lines(name_blob, name_bytes, n_names, names))
      return ffx_fail(FFX_BAD_ARG, "a buffer does not hold the announced number of strings");
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  }
  for (int64_t p = 0; p < n_pairs; ++p)
    if ((uint32_t)id_codes[p] >= (uint64_t)n_names) return ffx_fail(FFX_BAD_ARG, "row %lld holds id code %d, outside the %lld names", (long long)p, id_codes[p], (long long)n_names);
  return save_rows(path, q_names, n_queries, pair_off, n_pairs, [&](int64_t p) { return names[(size_t)id_codes[p]]; }, scores, tag);
}

}  // extern "C"
