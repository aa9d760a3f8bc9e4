// run_file.cpp -- TREC run files: parsed straight into the CSR the kernels take, and written back.
//
// Replaces pandas.read_csv + the Ranking constructor (src/fast_forward/ranking.py:388-409, :67-121) and Ranking.save's
// to_csv (ranking.py:348-366) on the file-to-file path: a run of 34.9 M lines is 1.4 GB of text each way, and every real
// call of the re-ranking path starts from a run the index has never seen.  Everything here is by host threads over pieces
// of the text:
//   reading    the file is mapped: the threads parse the page cache in place (no copy of the text, no fresh page)
//   parsing    pieces cut at line ends; per line six fields, the score through an exact decimal fast path, the candidate id
//              either looked up in the index's id table (look-ups of eight lines in flight at a time: each is three dependent
//              cache misses) or hashed for the numbering that follows; query ids are run-length encoded per piece
//   grouping   every piece writes its rows where they stand if the file is grouped by query and drops no row (its lines were
//              counted first); otherwise the rows are copied run by run to where their query's block stands (order of first
//              appearance of the query, file order inside)
//   numbering  without an id table the candidate ids are numbered by string_code.h (codes + the distinct names)
//   writing    batches of lines are formatted by threads into their own buffers and written with pwrite at the offset the
//              batches before them end at (a chain of batch ends, published in order): formatting and the copies into the
//              page cache all run in parallel
// No device code here; plain C++17.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cctype>
#include <charconv>
#include <cmath>
#include <numeric>
#include <system_error>

#include "host_util.h"
#include "string_code.h"

using namespace ffx_host;

// one parsed TREC run: rows grouped by query (order of first appearance), rows of a query in file order
struct ffx_run {
  ffx_ids qids;                   // query ids in order of first appearance
  std::vector<int64_t> pair_off;  // [Q+1]
  Buf<int32_t> pair_ids;          // ordinal in the id table given to ffx_run_open, or (opened without one) code into `names`
  Buf<float> score;
  int64_t n_pairs = 0;
  std::string tag;               // last column of the first line (Ranking.name, ranking.py:409)
  const ffx_ids* ids = nullptr;  // borrowed: candidate id strings for writing
  // opened without an id table (Ranking.from_file): the distinct candidate ids, numbered by string_code.h
  char* names = nullptr;  // joined by '\n'
  int64_t names_bytes = 0, n_names = 0;
  Buf<int64_t> name_start;  // [n_names + 1]: name c is names[name_start[c] .. name_start[c+1] - 1)
  int32_t q_flags = 0, id_flags = 0, misc_flags = 0;
  ~ffx_run() { free(names); }
  std::string_view name(int32_t c) const { return {names + name_start[(size_t)c], (size_t)(name_start[(size_t)c + 1] - 1 - name_start[(size_t)c])}; }
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

// [+-]digits[.digits][e[+-]digits] with at most 19 significant digits, a significand below 2^53 and a decimal exponent
// within +-22: the value is significand * 10^e or significand / 10^-e, ONE correctly rounded IEEE operation on two exact
// operands -- the double strtod / from_chars return (Clinger's fast path).  Every score a run file usually holds
// ("12.345678") takes this road, at a fraction of from_chars' cost; anything else returns false and the general code decides.
inline bool decimal_fast(const char* p, const char* e, double* out) {
  static const double pow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                   1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  bool neg = false;
  if (p < e && (*p == '-' || *p == '+')) neg = *p++ == '-';
  uint64_t m = 0;
  int nd = 0, exp10 = 0;
  bool any = false;
  for (; p < e && *p >= '0' && *p <= '9'; ++p) {
    any = true;
    if (nd >= 19) return false;
    m = m * 10 + (uint64_t)(*p - '0');
    nd += m != 0;
  }
  if (p < e && *p == '.') {
    for (++p; p < e && *p >= '0' && *p <= '9'; ++p) {
      any = true;
      if (nd >= 19) return false;
      m = m * 10 + (uint64_t)(*p - '0');
      nd += m != 0;
      --exp10;
    }
  }
  if (!any) return false;
  if (p < e && (*p == 'e' || *p == 'E')) {
    ++p;
    bool eneg = false;
    if (p < e && (*p == '-' || *p == '+')) eneg = *p++ == '-';
    if (p >= e) return false;
    int ex = 0;
    for (; p < e && *p >= '0' && *p <= '9'; ++p) {
      ex = ex * 10 + (*p - '0');
      if (ex > 10000) return false;
    }
    exp10 += eneg ? -ex : ex;
  }
  if (p != e) return false;
  if (m == 0) {
    *out = neg ? -0.0 : 0.0;
    return true;
  }
  if (m > (1ull << 53) || exp10 < -22 || exp10 > 22) return false;
  double v = (double)m;
  v = exp10 < 0 ? v / pow10[-exp10] : v * pow10[exp10];
  *out = neg ? -v : v;
  return true;
}

// the general road: 0 = *out holds the value, 1 = a missing score (the row is dropped), 2 = not a score pandas would read
inline int parse_score_slow(std::string_view field, double* out) {
  const int kind = score_token_kind(field);
  if (kind != 0) return kind;
  // correctly rounded double (what read_csv gives for these spellings); from_chars needs no terminator and takes no leading '+'
  const char* s0 = field.data();
  const char* s1 = s0 + field.size();
  if (s0 < s1 && *s0 == '+') ++s0;
  const auto res = std::from_chars(s0, s1, *out);
  if (res.ec == std::errc::result_out_of_range) {  // strtod's answer for these: +-inf / +-0
    const std::string tmp(field);
    char* endp = nullptr;
    *out = strtod(tmp.c_str(), &endp);
    return endp == tmp.c_str() + tmp.size() ? 0 : 2;
  }
  return (res.ec == std::errc() && res.ptr == s1) ? 0 : 2;
}

// ---- formatting --------------------------------------------------------------------------------------------
// shortest round-trip decimal of a float, laid out like numpy's str(np.float32), which is what pandas.to_csv writes for a
// float32 column (ranking.py:360-366).  At most 24 bytes at w; returns the end.
inline char* write_float_repr(char* w, float v) {
  if (std::isnan(v)) {
    memcpy(w, "nan", 3);
    return w + 3;
  }
  if (std::isinf(v)) {
    if (v < 0) *w++ = '-';
    memcpy(w, "inf", 3);
    return w + 3;
  }
  char buf[32];
  const auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);  // [-]d[.ddd]e[+-]XX
  const char* s = buf;
  const char* end = r.ptr;
  if (*s == '-') {
    *w++ = '-';
    ++s;
  }
  char digits[16];
  int nd = 0;
  const char* c = s;
  for (; c < end && *c != 'e'; ++c)
    if (*c != '.') digits[nd++] = *c;
  int exp10 = 0;
  {
    const char* x = c + 1;
    bool eneg = false;
    if (x < end && (*x == '-' || *x == '+')) eneg = *x++ == '-';
    for (; x < end; ++x) exp10 = exp10 * 10 + (*x - '0');
    if (eneg) exp10 = -exp10;
  }
  // numpy switches to scientific on the VALUE (|v| < 1e-4 or >= 1e16), not on the decimal exponent
  // of the shortest digits: float32(1e-4) = 9.99999975e-05 prints as 1e-04
  const double mag = std::fabs((double)v);
  if ((mag != 0 && mag < 1e-4) || mag >= 1e16) {
    *w++ = digits[0];
    if (nd > 1) {
      *w++ = '.';
      memcpy(w, digits + 1, (size_t)nd - 1);
      w += nd - 1;
    }
    *w++ = 'e';
    *w++ = exp10 < 0 ? '-' : '+';
    const int a = std::abs(exp10) % 100;  // a float's decimal exponent has two digits
    *w++ = (char)('0' + a / 10);
    *w++ = (char)('0' + a % 10);
  } else if (exp10 < 0) {
    *w++ = '0';
    *w++ = '.';
    for (int z = 0; z < -exp10 - 1; ++z) *w++ = '0';
    memcpy(w, digits, (size_t)nd);
    w += nd;
  } else {
    const int int_len = exp10 + 1;
    if (nd <= int_len) {
      memcpy(w, digits, (size_t)nd);
      w += nd;
      for (int z = 0; z < int_len - nd; ++z) *w++ = '0';
      *w++ = '.';
      *w++ = '0';
    } else {
      memcpy(w, digits, (size_t)int_len);
      w += int_len;
      *w++ = '.';
      memcpy(w, digits + int_len, (size_t)(nd - int_len));
      w += nd - int_len;
    }
  }
  return w;
}

inline char* write_uint(char* w, uint64_t v) {
  char tmp[24];
  int n = 0;
  do {
    tmp[n++] = (char)('0' + v % 10);
    v /= 10;
  } while (v);
  while (n) *w++ = tmp[--n];
  return w;
}

// a growing byte buffer the formatters write through a bare pointer
struct Out {
  Buf<char> mem;
  char* w = nullptr;
  void clear() { w = mem.data(); }
  size_t size() const { return (size_t)(w - mem.data()); }
  void room(size_t extra) {  // at least `extra` more bytes may be written at w
    if (mem.data() && size() + extra <= mem.size()) return;
    const size_t have = mem.data() ? size() : 0;
    size_t cap = std::max<size_t>(mem.size() * 2, 1 << 16);
    while (cap < have + extra) cap *= 2;
    Buf<char> bigger(cap);
    if (have) memcpy(bigger.data(), mem.data(), have);
    mem = std::move(bigger);
    w = mem.data() + have;
  }
};

// The lines of a run, n_rows of them in n_groups groups (queries; group g holds rows group_off[g] .. group_off[g+1] of the
// OUTPUT), written to `path`.  format(g, r_lo, r_hi, out) appends the lines of the group's rows r_lo .. r_hi-1 (0-based inside
// the group: rank r + 1).  Batches of ~32 K rows -- cut at any row, so one long query does not become one thread's job -- go
// round the threads; a thread formats a batch into its own buffer, learns where the batch before it ends (end_of[b - 1],
// published by whoever formatted that one), publishes its own end and writes with pwrite.  Every thread takes its batches
// in ascending order, so the chain of ends is always being extended by somebody and nobody waits on a batch behind its own.
template <class Format>
int write_lines(const char* path, int64_t n_groups, const int64_t* group_off, Format&& format, StageClock* clock) {
  const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
  if (fd < 0) return ffx_fail(FFX_BAD_ARG, "cannot create %s", path);
  const int64_t n_rows = n_groups > 0 ? group_off[n_groups] : 0;
  constexpr int64_t kBatch = 1 << 15;
  const int64_t n_batches = (n_rows + kBatch - 1) / kBatch;
  const int T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::min(64, usable_cores()), n_batches}));
  std::vector<std::atomic<int64_t>> end_of((size_t)n_batches + 1);
  for (auto& e : end_of) e.store(-1, std::memory_order_relaxed);
  end_of[0].store(0);  // end_of[b + 1] = where batch b ends in the file
  std::atomic<bool> failed(false), oom(false);
  auto work = [&](int t) {
    Out out;
    for (int64_t b = t; b < n_batches; b += T) {
      const int64_t lo = b * kBatch, hi = std::min(n_rows, lo + kBatch);
      bool formatted = false;
      if (!failed.load(std::memory_order_relaxed) && !oom.load(std::memory_order_relaxed)) {
        try {
          out.room(1);
          out.clear();
          int64_t g = std::upper_bound(group_off, group_off + n_groups + 1, lo) - group_off - 1;
          for (int64_t r = lo; r < hi;) {
            while (group_off[g + 1] <= r) ++g;
            const int64_t stop = std::min(hi, group_off[g + 1]);
            format(g, r - group_off[g], stop - group_off[g], out);
            r = stop;
          }
          formatted = true;
        } catch (const std::bad_alloc&) {  // (an exception leaving a thread would end the process)
          oom.store(true);
        }
      }
      const int64_t bytes = formatted ? (int64_t)out.size() : 0;
      int64_t at;
      while ((at = end_of[(size_t)b].load(std::memory_order_acquire)) < 0) std::this_thread::yield();
      end_of[(size_t)b + 1].store(at + bytes, std::memory_order_release);
      for (int64_t done = 0; done < bytes && !failed.load(std::memory_order_relaxed);) {
        const ssize_t k = pwrite(fd, out.mem.data() + done, (size_t)(bytes - done), (off_t)(at + done));
        if (k <= 0) {
          failed.store(true);
          break;
        }
        done += k;
      }
    }
  };
  try {
    run_threads(T, work);
  } catch (const std::system_error&) {
    close(fd);
    return ffx_fail(FFX_OOM, "could not start host threads while writing %s", path);
  }
  if (clock) clock->lap("format_and_write");
  const bool bad_close = close(fd) != 0;
  if (oom.load()) return ffx_fail(FFX_OOM, "host allocation failed while writing %s", path);
  if (failed.load() || bad_close) return ffx_fail(FFX_BAD_ARG, "write to %s failed", path);
  return FFX_OK;
}

// one line "q_id\tQ0\tid\trank\tscore\ttag\n" at out.w
inline void write_line(Out& out, std::string_view qid, std::string_view id, int64_t rank, float score, std::string_view tag) {
  out.room(qid.size() + id.size() + tag.size() + 64);
  char* w = out.w;
  memcpy(w, qid.data(), qid.size());
  w += qid.size();
  memcpy(w, "\tQ0\t", 4);
  w += 4;
  memcpy(w, id.data(), id.size());
  w += id.size();
  *w++ = '\t';
  w = write_uint(w, (uint64_t)rank);
  *w++ = '\t';
  w = write_float_repr(w, score);
  *w++ = '\t';
  memcpy(w, tag.data(), tag.size());
  w += tag.size();
  *w++ = '\n';
  out.w = w;
}

// where every one of the n strings of a buffer starts when the byte '\n' stands between them: start[i] .. start[i+1] - 1 are
// the bytes of string i (start[n] = bytes + 1).  By threads over byte ranges (separators counted, then written).  False if
// the buffer does not hold exactly n strings.
bool line_starts(const char* blob, int64_t bytes, int64_t n, Buf<int64_t>& start) {
  start.alloc((size_t)n + 1);
  if (n > 0 && !blob) return false;
  if (n == 0) {
    start[0] = 1;
    return bytes == 0;
  }
  const int T = threads_for(bytes, 1 << 20);
  std::vector<int64_t> seps((size_t)T + 1, 0);
  auto range = [&](int t) { return std::pair<int64_t, int64_t>(bytes * t / T, bytes * (t + 1) / T); };
  run_threads(T, [&](int t) {
    const auto [lo, hi] = range(t);
    int64_t c = 0;
    for (const char* p = blob + lo; (p = static_cast<const char*>(memchr(p, '\n', (size_t)(blob + hi - p)))) != nullptr; ++p) ++c;
    seps[(size_t)t + 1] = c;
  });
  for (int t = 0; t < T; ++t) seps[(size_t)t + 1] += seps[(size_t)t];
  if (seps[(size_t)T] != n - 1) return false;
  start[0] = 0;
  start[(size_t)n] = bytes + 1;
  run_threads(T, [&](int t) {
    const auto [lo, hi] = range(t);
    int64_t i = seps[(size_t)t] + 1;
    for (const char* p = blob + lo; (p = static_cast<const char*>(memchr(p, '\n', (size_t)(blob + hi - p)))) != nullptr; ++p) start[(size_t)i++] = (p - blob) + 1;
  });
  return true;
}

// one piece of the text and what parsing it gave.  A piece writes its rows into the run's arrays from row `base` on -- it
// was given room for as many rows as it has lines --, so a file that is grouped by query and drops no row (the usual one)
// is in place when the last piece is done.
struct QRun {
  const char* p;
  uint32_t len;
  int64_t rows;
};
struct Piece {
  const char *begin = nullptr, *end = nullptr;
  int64_t base = 0, room = 0, rows = 0;
  std::vector<QRun> q_runs;  // (query id, consecutive rows)
  int32_t q_flags = 0, id_flags = 0, misc_flags = 0;
  std::string_view tag;
  bool has_tag = false;
  int64_t lines = 0;
  int err = FFX_OK;  // first problem of the piece
  int64_t err_line = 0;
  std::string err_text;
  bool oom = false;
};

inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r'; }

// score[] and, with an id table, id[] (ordinals) or, without, ref[] ((offset of the id in the text) << 16 | length) are the
// run's arrays
void parse_piece(Piece& pc, const char* text_base, const ffx_ids* ids, float* score, int32_t* id, uint64_t* ref) {
  const bool standalone = ids == nullptr;
  constexpr int B = 8;  // look-ups in flight
  std::string_view pend_sv[B];
  uint64_t pend_h[B];
  int64_t pend_line[B];
  int n_pend = 0;
  int64_t w = pc.base;      // next row
  int64_t w_id = pc.base;   // next resolved id (trails w by the look-ups in flight)
  auto fail = [&](int code, int64_t line, std::string text) {
    if (pc.err == FFX_OK || line < pc.err_line) pc.err = code, pc.err_line = line, pc.err_text = std::move(text);
  };
  auto flush = [&]() {  // the pending look-ups: slot, offsets and bytes of the eight candidates fetched side by side
    for (int b = 0; b < n_pend; ++b) {
      const uint64_t v = ids->slots[pend_h[b] & ids->mask];
      if (v != 0) __builtin_prefetch(&ids->off[(size_t)((uint32_t)v - 1)]);
    }
    for (int b = 0; b < n_pend; ++b) {
      const uint64_t v = ids->slots[pend_h[b] & ids->mask];
      if (v != 0) __builtin_prefetch(ids->arena.data() + ids->off[(size_t)((uint32_t)v - 1)]);
    }
    for (int b = 0; b < n_pend; ++b) {
      const int32_t o = ids->find_hashed(pend_sv[b], pend_h[b]);
      if (o < 0) {
        fail(FFX_NOT_FOUND, pend_line[b], std::string(pend_sv[b]));
        n_pend = 0;
        return false;
      }
      id[w_id++] = o;
    }
    n_pend = 0;
    return true;
  };
  try {
    const char* p = pc.begin;
    const char* const end = pc.end;
    if (standalone && end > p && memchr(p, '"', (size_t)(end - p))) pc.misc_flags |= FFX_RUN_QUOTES;
    std::string_view last_q;
    while (p < end) {
      // the fields of one line
      std::string_view field[6];
      int nf = 0;
      const char* c = p;
      for (;;) {
        while (c < end && is_blank(*c)) ++c;
        if (c >= end || *c == '\n') break;
        const char* s0 = c;
        while (c < end && !is_blank(*c) && *c != '\n') ++c;
        if (nf < 6) field[nf] = {s0, (size_t)(c - s0)};
        ++nf;
      }
      p = c < end ? c + 1 : end;
      ++pc.lines;
      if (nf == 0) continue;  // blank line
      if (nf < 5) {
        fail(FFX_BAD_ARG, pc.lines, "expected 'q_id Q0 id rank score [name]'");
        break;
      }
      double sc = 0.0;
      if (!decimal_fast(field[4].data(), field[4].data() + field[4].size(), &sc)) {
        const int kind = parse_score_slow(field[4], &sc);
        if (kind == 2) {
          fail(FFX_BAD_ARG, pc.lines, "bad score '" + std::string(field[4]) + "'");
          break;
        }
        if (kind == 1) continue;  // a missing score: the reference drops the row (ranking.py:103)
      }
      const bool new_run = pc.q_runs.empty() || last_q != field[0];
      if (standalone) {
        if (field[2].size() > 0xFFFFu) {
          fail(FFX_BAD_ARG, pc.lines, "a candidate id of more than 65535 bytes");
          break;
        }
        pc.id_flags |= token_class(field[2]);
        if (new_run) pc.q_flags |= token_class(field[0]);
        if (nf < 6) pc.misc_flags |= FFX_RUN_SHORT_LINE;
        if (nf > 6) pc.misc_flags |= FFX_RUN_LONG_LINE;
        ref[w] = ((uint64_t)(field[2].data() - text_base) << 16) | (uint64_t)field[2].size();
      } else {
        pend_sv[n_pend] = field[2];
        pend_h[n_pend] = fnv1a(field[2].data(), field[2].size());
        pend_line[n_pend] = pc.lines;
        __builtin_prefetch(&ids->slots[pend_h[n_pend] & ids->mask]);
        if (++n_pend == B && !flush()) break;
      }
      if (!pc.has_tag && nf >= 6) pc.tag = field[5], pc.has_tag = true;
      if (new_run) {
        pc.q_runs.push_back({field[0].data(), (uint32_t)field[0].size(), 0});
        last_q = field[0];
      }
      ++pc.q_runs.back().rows;
      score[w++] = (float)sc;  // scores become float32 (Ranking dtype, ranking.py:72)
    }
    if (!standalone && n_pend > 0) flush();  // (also after a parse error: an unknown id on an earlier line comes first)
    pc.rows = w - pc.base;
  } catch (const std::bad_alloc&) {
    pc.oom = true;
  }
}

// a file mapped for reading (no copy of the text, no fresh pages: the threads read the page cache), or -- where a mapping
// cannot be had -- read into memory by all threads at once
struct Text {
  const char* data = nullptr;
  int64_t size = 0;
  void* mapped = nullptr;
  Buf<char> owned;
  ~Text() {
    if (mapped) munmap(mapped, (size_t)size);
  }
};

}  // namespace

extern "C" {

// Line format (Ranking.from_file, ranking.py:400-409): q_id Q0 id rank score name, any whitespace.
// Candidate ids are resolved against `ids` while parsing; the first unknown id is FFX_NOT_FOUND with
// its 1-based line number in *bad_line (the reference's IndexError, index/util.py:38-39).
// A (query, id) pair that appears twice is FFX_BAD_ARG (ValueError, ranking.py:95-98).
int ffx_run_open(const char* path, const ffx_ids* ids, ffx_run** out, int64_t* bad_line) {
  if (bad_line) *bad_line = -1;
  if (!path || !out) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  *out = nullptr;
  const bool standalone = ids == nullptr;  // number the candidate ids instead of resolving them
  StageClock clock;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) return ffx_fail(FFX_BAD_ARG, "cannot open %s", path);
  struct Closer {  // (the descriptor is closed on every way out)
    int fd;
    ~Closer() { close(fd); }
  } closer{fd};
  struct stat st;
  if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return ffx_fail(FFX_BAD_ARG, "%s is not a regular file", path);
  const int64_t size = (int64_t)st.st_size;
  ffx_run* run = nullptr;
  int rc = FFX_OK;
  try {
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::min(64, usable_cores()), size >> 20}));
    Text text;
    text.size = size;
    if (size > 0) {
      void* m = mmap(nullptr, (size_t)size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (m != MAP_FAILED) {
        text.mapped = m;
        text.data = static_cast<const char*>(m);
      } else {
        text.owned.alloc((size_t)size);
        std::atomic<bool> short_read(false);
        run_threads(T, [&](int t) {
          const int64_t lo = size * t / T, hi = size * (t + 1) / T;
          for (int64_t at = lo; at < hi;) {
            const ssize_t k = pread(fd, text.owned.data() + at, (size_t)std::min<int64_t>(hi - at, 1 << 26), (off_t)at);
            if (k <= 0) {
              short_read.store(true);
              return;
            }
            at += k;
          }
        });
        if (short_read.load()) return ffx_fail(FFX_BAD_ARG, "short read on %s", path);
        text.data = text.owned.data();
      }
    }
    run = new ffx_run();
    run->ids = ids;
    const char* const base = text.data;
    const char* const end = base + size;
    // pieces cut at line ends, a few per thread (lines differ in cost: missing scores, long ids); their lines counted: a
    // piece has room for as many rows as it has lines
    const int n_pieces = (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)T * 4, size >> 18));
    std::vector<Piece> pieces((size_t)n_pieces);
    {
      const char* cut = base;
      for (int k = 0; k < n_pieces; ++k) {
        pieces[(size_t)k].begin = cut;
        const char* want = k + 1 == n_pieces ? end : base + size * (k + 1) / n_pieces;
        if (want < cut) want = cut;
        if (want < end) {
          const char* nl = (const char*)memchr(want, '\n', (size_t)(end - want));
          want = nl ? nl + 1 : end;
        }
        pieces[(size_t)k].end = cut = want;
      }
    }
    std::atomic<int> next_piece(0);
    run_threads(T, [&](int) {
      for (;;) {
        const int k = next_piece.fetch_add(1);
        if (k >= n_pieces) return;
        Piece& pc = pieces[(size_t)k];
        int64_t c = 0;
        for (const char* p = pc.begin; p < pc.end && (p = static_cast<const char*>(memchr(p, '\n', (size_t)(pc.end - p)))) != nullptr; ++p) ++c;
        pc.room = c + ((pc.end > pc.begin && pc.end[-1] != '\n') ? 1 : 0);  // (the last line of the file need not end in a line break)
      }
    });
    int64_t room = 0;
    for (Piece& pc : pieces) {
      pc.base = room;
      room += pc.room;
    }
    clock.lap("map_and_count_lines");
    if (room >= INT32_MAX - 1) {
      delete run;
      return ffx_fail(FFX_BAD_ARG, "%s: too many rows for 32-bit positions", path);
    }
    run->score.alloc((size_t)room);
    run->pair_ids.alloc((size_t)room);
    Buf<uint64_t> ref;
    if (standalone) ref.alloc((size_t)room);
    next_piece.store(0);
    std::atomic<int> first_bad(INT32_MAX);  // pieces behind a faulty one need not be parsed
    run_threads(T, [&](int) {
      for (;;) {
        const int k = next_piece.fetch_add(1);
        if (k >= n_pieces) return;
        if (k > first_bad.load(std::memory_order_relaxed)) continue;
        parse_piece(pieces[(size_t)k], base, ids, run->score.data(), run->pair_ids.data(), ref.data());
        if (pieces[(size_t)k].err != FFX_OK || pieces[(size_t)k].oom) {
          int cur = first_bad.load();
          while (k < cur && !first_bad.compare_exchange_weak(cur, k)) {
          }
        }
      }
    });
    clock.lap("parse");
    // problems, in file order
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
    }
    if (rc == FFX_OK) {
      // the queries numbered in order of first appearance; every run of rows learns where its query's block takes it
      struct Move {
        int piece;
        int64_t src, rows, dst;
        int32_t q;
      };
      std::vector<Move> moves;
      std::vector<int64_t> rows_of;  // per query
      int64_t n = 0;
      for (int k = 0; k < n_pieces; ++k) {
        Piece& pc = pieces[(size_t)k];
        if (run->tag.empty() && pc.has_tag) run->tag.assign(pc.tag);
        run->q_flags |= pc.q_flags;
        run->id_flags |= pc.id_flags;
        run->misc_flags |= pc.misc_flags;
        int64_t src = pc.base;
        for (const QRun& qr : pc.q_runs) {
          if (run->qids.size() >= INT32_MAX - 1) throw std::bad_alloc();
          const int32_t q = run->qids.intern(std::string_view(qr.p, qr.len));
          if ((size_t)q >= rows_of.size()) rows_of.resize((size_t)q + 1, 0);
          if (!moves.empty() && moves.back().q == q && moves.back().piece == k) {
            moves.back().rows += qr.rows;
          } else {
            moves.push_back({k, src, qr.rows, 0, q});
          }
          rows_of[(size_t)q] += qr.rows;
          src += qr.rows;
          n += qr.rows;
        }
      }
      clock.lap("group.queries_numbered");
      const int64_t nq = run->qids.size();
      run->pair_off.assign((size_t)nq + 1, 0);
      for (int64_t q = 0; q < nq; ++q) run->pair_off[(size_t)q + 1] = run->pair_off[(size_t)q] + rows_of[(size_t)q];
      bool in_place = true;
      {
        std::vector<int64_t> cursor(run->pair_off.begin(), run->pair_off.end() - 1);
        for (Move& mv : moves) {  // stable: rows of a query keep file order
          mv.dst = cursor[(size_t)mv.q];
          cursor[(size_t)mv.q] += mv.rows;
          in_place = in_place && mv.dst == mv.src;
        }
      }
      run->n_pairs = n;
      if (!in_place) {  // dropped rows, or a query whose rows stand apart: the rows are moved into fresh arrays, by pieces
        Buf<float> score2((size_t)n);
        Buf<int32_t> ids2((size_t)n);
        Buf<uint64_t> ref2;
        if (standalone) ref2.alloc((size_t)n);
        std::vector<size_t> first_move((size_t)n_pieces + 1, moves.size());  // (moves stand in piece order)
        {
          size_t m = 0;
          for (int k = 0; k <= n_pieces; ++k) {
            while (m < moves.size() && moves[m].piece < k) ++m;
            first_move[(size_t)k] = m;
          }
        }
        next_piece.store(0);
        run_threads(T, [&](int) {
          for (;;) {
            const int k = next_piece.fetch_add(1);
            if (k >= n_pieces) return;
            for (size_t m = first_move[(size_t)k]; m < first_move[(size_t)k + 1]; ++m) {
              const Move& mv = moves[m];
              memcpy(score2.data() + mv.dst, run->score.data() + mv.src, (size_t)mv.rows * sizeof(float));
              if (standalone)
                memcpy(ref2.data() + mv.dst, ref.data() + mv.src, (size_t)mv.rows * sizeof(uint64_t));
              else
                memcpy(ids2.data() + mv.dst, run->pair_ids.data() + mv.src, (size_t)mv.rows * sizeof(int32_t));
            }
          }
        });
        run->score = std::move(score2);
        run->pair_ids = std::move(ids2);
        if (standalone) ref = std::move(ref2);
      }
      clock.lap("group.rows_moved");
      int64_t n_codes = ids ? ids->size() : 0;
      if (standalone) {
        const int TC = threads_for(n, 1 << 16);
        auto view = [&](int64_t i) { return std::string_view(base + (ref[(size_t)i] >> 16), (size_t)(ref[(size_t)i] & 0xFFFFu)); };
        rc = strings_code_views(n, TC, view, '\n', run->pair_ids.data(), &run->names, &run->names_bytes, &run->n_names, &clock, &run->name_start);
        n_codes = run->n_names;
      }
      // duplicate (query, id) pairs
      if (rc == FFX_OK && n > 0 && first_group_with_repeated_code(run->pair_ids.data(), run->pair_off.data(), nq, n_codes, threads_for(n, 1 << 16)) >= 0)
        rc = ffx_fail(FFX_BAD_ARG, "Only one score per query-document/passage pair is allowed.");
      clock.lap("repeated_pairs");
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
  if (n_pairs) *n_pairs = run->n_pairs;
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
// (Ranking.cut, ranking.py:279-291: order by score descending, ties in file order).  A run opened without an id table
// hands out its id CODES (ffx_run_names) where the other kind hands out ordinals.
int ffx_run_export(ffx_run* run, int64_t cutoff, int64_t* pair_off, int32_t* pair_ids, float* sparse, int64_t* n_pairs_out) {
  if (!run || !pair_off) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  const int64_t nq = run->qids.size();
  if (cutoff > 0) {  // trim in place, once; queries by threads
    try {
      std::vector<int64_t> new_off((size_t)nq + 1, 0);
      for (int64_t q = 0; q < nq; ++q) new_off[(size_t)q + 1] = new_off[(size_t)q] + std::min(cutoff, run->pair_off[(size_t)q + 1] - run->pair_off[(size_t)q]);
      const int64_t kept = new_off[(size_t)nq];
      if (kept < run->n_pairs) {
        Buf<int32_t> ids2((size_t)kept);
        Buf<float> sc2((size_t)kept);
        std::atomic<int64_t> next(0);
        std::atomic<bool> oom(false);
        run_threads(threads_for(run->n_pairs, 1 << 16), [&](int) {
          try {
            std::vector<int64_t> idx;
            for (;;) {
              const int64_t q = next.fetch_add(1);
              if (q >= nq) return;
              const int64_t lo = run->pair_off[(size_t)q], hi = run->pair_off[(size_t)q + 1];
              int64_t w = new_off[(size_t)q];
              if (hi - lo <= cutoff) {
                for (int64_t i = lo; i < hi; ++i, ++w) ids2[(size_t)w] = run->pair_ids[(size_t)i], sc2[(size_t)w] = run->score[(size_t)i];
                continue;
              }
              idx.resize((size_t)(hi - lo));
              std::iota(idx.begin(), idx.end(), lo);
              std::stable_sort(idx.begin(), idx.end(), [&](int64_t a, int64_t b) { return run->score[(size_t)a] > run->score[(size_t)b]; });
              idx.resize((size_t)cutoff);
              std::sort(idx.begin(), idx.end());  // survivors back in file order
              for (int64_t i : idx) ids2[(size_t)w] = run->pair_ids[(size_t)i], sc2[(size_t)w] = run->score[(size_t)i], ++w;
            }
          } catch (const std::bad_alloc&) {
            oom.store(true);
          }
        });
        if (oom.load()) throw std::bad_alloc();
        run->pair_off.swap(new_off);
        run->pair_ids = std::move(ids2);
        run->score = std::move(sc2);
        run->n_pairs = kept;
      }
    } catch (const std::bad_alloc&) {
      return ffx_fail(FFX_OOM, "host allocation failed");
    } catch (const std::system_error&) {
      return ffx_fail(FFX_OOM, "could not start host threads");
    }
  }
  const int64_t n = run->n_pairs;
  if (n_pairs_out) *n_pairs_out = n;
  memcpy(pair_off, run->pair_off.data(), (size_t)(nq + 1) * sizeof(int64_t));
  if (n > 0 && (pair_ids || sparse)) {
    const int T = threads_for(n, 1 << 20);
    try {
      run_threads(T, [&](int t) {
        const int64_t lo = n * t / T, hi = n * (t + 1) / T;
        if (pair_ids) memcpy(pair_ids + lo, run->pair_ids.data() + lo, (size_t)(hi - lo) * sizeof(int32_t));
        if (sparse) memcpy(sparse + lo, run->score.data() + lo, (size_t)(hi - lo) * sizeof(float));
      });
    } catch (const std::system_error&) {
      return ffx_fail(FFX_OOM, "could not start host threads");
    }
  }
  return FFX_OK;
}

// standalone runs (opened without an id table): what pandas would make of the q_id / id columns
// (FFX_TOKEN_* classes OR-ed over the column) and of the file's shape (FFX_RUN_* bits)
int ffx_run_flags(const ffx_run* run, int32_t* q_flags, int32_t* id_flags, int32_t* misc_flags) {
  if (!run) return ffx_fail(FFX_BAD_ARG, "NULL run");
  if (q_flags) *q_flags = run->q_flags;
  if (id_flags) *id_flags = run->id_flags;
  if (misc_flags) *misc_flags = run->misc_flags;
  return FFX_OK;
}

// standalone runs: the DISTINCT candidate ids in code order, joined by '\n' (a view into the run: valid until ffx_run_close)
int ffx_run_names(const ffx_run* run, const char** blob, int64_t* bytes, int64_t* n_names) {
  if (!run || !blob || !bytes || !n_names) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  if (run->ids) return ffx_fail(FFX_BAD_ARG, "this run was resolved against an id table: it holds ordinals, not codes");
  *blob = run->names;
  *bytes = run->names_bytes;
  *n_names = run->n_names;
  return FFX_OK;
}

// standalone runs: the candidate ids in CSR order (grouped by query, file order inside), joined by
// '\n'.  Call with buf == NULL to learn the size.
int ffx_run_ids_blob(const ffx_run* run, char* buf, int64_t cap, int64_t* needed) {
  if (!run || !needed) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  if (run->ids) return ffx_fail(FFX_BAD_ARG, "this run was resolved against an id table: it holds ordinals, not strings");
  int64_t total = 0;
  for (int64_t i = 0; i < run->n_pairs; ++i) total += (int64_t)run->name(run->pair_ids[(size_t)i]).size() + 1;
  if (total > 0) --total;  // no separator after the last one
  *needed = total;
  if (!buf) return FFX_OK;
  if (cap < total) return ffx_fail(FFX_BAD_ARG, "buffer too small");
  char* w = buf;
  for (int64_t i = 0; i < run->n_pairs; ++i) {
    if (i) *w++ = '\n';
    const std::string_view v = run->name(run->pair_ids[(size_t)i]);
    memcpy(w, v.data(), v.size());
    w += v.size();
  }
  return FFX_OK;
}

// write `q_id Q0 id rank score tag` (tab separated; Ranking.save, ranking.py:348-366): queries in
// descending q_id order, candidates of a query in the order given by `order` (positions into the
// CSR, e.g. the device sort of ffx_plan_run); keep > 0 writes only the first `keep` of each query.
int ffx_run_write(const ffx_run* run, const char* path, const float* scores, const int32_t* order, int64_t keep,
                  const char* tag) {
  if (!run || !path || !scores || !order) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  StageClock clock;
  try {
    const int64_t nq = run->qids.size();
    std::vector<int64_t> qorder((size_t)nq);
    std::iota(qorder.begin(), qorder.end(), 0);
    std::sort(qorder.begin(), qorder.end(), [&](int64_t a, int64_t b) { return run->qids.get((int32_t)a) > run->qids.get((int32_t)b); });
    std::vector<int64_t> out_off((size_t)nq + 1, 0);
    for (int64_t qi = 0; qi < nq; ++qi) {
      const int64_t q = qorder[(size_t)qi], n = run->pair_off[(size_t)q + 1] - run->pair_off[(size_t)q];
      out_off[(size_t)qi + 1] = out_off[(size_t)qi] + ((keep > 0 && n > keep) ? keep : n);
    }
    const int64_t n_pairs = run->n_pairs;
    std::atomic<bool> bad_order(false);
    const std::string name = tag ? tag : run->tag;
    auto id_of = [&](int64_t p) { return run->pair_ids[(size_t)p]; };
    auto format = [&](int64_t qi, int64_t r_lo, int64_t r_hi, Out& out) {
      const int64_t q = qorder[(size_t)qi];
      const std::string_view qid = run->qids.get((int32_t)q);
      const int64_t lo = run->pair_off[(size_t)q];
      auto in_run = [&](int64_t p) { return p >= 0 && p < n_pairs; };
      for (int64_t r = r_lo; r < r_hi; ++r) {
        const int64_t p = order[lo + r];
        if (!in_run(p)) {
          bad_order.store(true);
          return;
        }
        // a row's line is a chain of dependent cache misses (its position in the run -> its ordinal and score -> where the
        // id's bytes start -> the bytes): each link is asked for a few rows ahead of the row being printed
        if (r + 24 < r_hi && in_run(order[lo + r + 24])) {
          __builtin_prefetch(&run->pair_ids[(size_t)order[lo + r + 24]]);
          __builtin_prefetch(&scores[order[lo + r + 24]]);
        }
        if (r + 16 < r_hi && in_run(order[lo + r + 16])) {
          const int32_t c = id_of(order[lo + r + 16]);
          if (run->ids) {
            if (c >= 0 && c < run->ids->size()) __builtin_prefetch(&run->ids->off[(size_t)c]);
          } else if (c >= 0 && c < run->n_names) {
            __builtin_prefetch(&run->name_start[(size_t)c]);
          }
        }
        if (r + 8 < r_hi && in_run(order[lo + r + 8])) {
          const int32_t c = id_of(order[lo + r + 8]);
          if (run->ids) {
            if (c >= 0 && c < run->ids->size()) __builtin_prefetch(run->ids->arena.data() + run->ids->off[(size_t)c]);
          } else if (c >= 0 && c < run->n_names) {
            __builtin_prefetch(run->names + run->name_start[(size_t)c]);
          }
        }
        const int32_t c = id_of(p);
        write_line(out, qid, run->ids ? run->ids->get(c) : run->name(c), r + 1, scores[p], name);
      }
    };
    const int rc = write_lines(path, nq, out_off.data(), format, &clock);
    if (rc == FFX_OK && bad_order.load()) return ffx_fail(FFX_BAD_ARG, "order holds a position outside the run");
    return rc;
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  }
}

}  // extern "C"

namespace {

int check_save_args(const char* path, const int64_t* pair_off, int64_t n_queries, int64_t n_pairs, const char* tag) {
  if (!path || !pair_off || n_queries < 0 || n_pairs < 0 || !tag) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (pair_off[0] != 0 || pair_off[n_queries] != n_pairs) return ffx_fail(FFX_BAD_ARG, "pair_off does not cover the rows");
  for (int64_t q = 0; q < n_queries; ++q)
    if (pair_off[q + 1] < pair_off[q]) return ffx_fail(FFX_BAD_ARG, "pair_off not monotone at query %lld", (long long)q);
  return FFX_OK;
}

inline std::string_view line_of(const char* blob, const Buf<int64_t>& start, int64_t i) {
  return {blob + start[(size_t)i], (size_t)(start[(size_t)i + 1] - 1 - start[(size_t)i])};
}

}  // namespace

extern "C" {

// Ranking.save (ranking.py:348-366) for a frame that is already grouped by query: the q_id and id
// columns arrive as two '\n'-joined buffers (n_queries names; n_pairs ids in frame order), pair_off
// gives every query's rows.  Lines "q_id\tQ0\tid\trank\tscore\tname", rank = 1-based position inside
// the query, score formatted like pandas' to_csv of a float32 column.  Formatted and written by threads.
int ffx_run_save(const char* path, const char* q_blob, int64_t q_bytes, int64_t n_queries, const int64_t* pair_off,
                 const char* id_blob, int64_t id_bytes, int64_t n_pairs, const float* scores, const char* tag) {
  if (int rc = check_save_args(path, pair_off, n_queries, n_pairs, tag)) return rc;
  if (n_pairs > 0 && (!id_blob || !scores || !q_blob)) return ffx_fail(FFX_BAD_ARG, "bad argument");
  StageClock clock;
  try {
    Buf<int64_t> q_start, id_start;
    if (!line_starts(q_blob, q_bytes, n_queries, q_start) || !line_starts(id_blob, id_bytes, n_pairs, id_start))
      return ffx_fail(FFX_BAD_ARG, "a buffer does not hold the announced number of strings");
    clock.lap("split_names");
    const std::string_view name(tag);
    auto format = [&](int64_t q, int64_t r_lo, int64_t r_hi, Out& out) {
      const std::string_view qid = line_of(q_blob, q_start, q);
      for (int64_t r = r_lo; r < r_hi; ++r) {
        const int64_t p = pair_off[q] + r;
        write_line(out, qid, line_of(id_blob, id_start, p), r + 1, scores[p], name);
      }
    };
    return write_lines(path, n_queries, pair_off, format, &clock);
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
}

// The same for rows whose candidate ids are CODES into a table of names (a run as the device hands it back:
// Ranking._assemble_coded): row p is written with name_blob's string number id_codes[p].
int ffx_run_save_coded(const char* path, const char* q_blob, int64_t q_bytes, int64_t n_queries, const int64_t* pair_off,
                       const char* name_blob, int64_t name_bytes, int64_t n_names, const int32_t* id_codes, int64_t n_pairs,
                       const float* scores, const char* tag) {
  if (int rc = check_save_args(path, pair_off, n_queries, n_pairs, tag)) return rc;
  if (n_names < 0 || (n_pairs > 0 && (!name_blob || !id_codes || !scores || !q_blob))) return ffx_fail(FFX_BAD_ARG, "bad argument");
  StageClock clock;
  try {
    Buf<int64_t> q_start, name_start;
    if (!line_starts(q_blob, q_bytes, n_queries, q_start) || !line_starts(name_blob, name_bytes, n_names, name_start))
      return ffx_fail(FFX_BAD_ARG, "a buffer does not hold the announced number of strings");
    clock.lap("split_names");
    {
      const int T = threads_for(n_pairs, 1 << 18);
      std::atomic<int64_t> bad(INT64_MAX);
      run_threads(T, [&](int t) {
        for (int64_t p = n_pairs * t / T; p < n_pairs * (t + 1) / T; ++p)
          if ((uint32_t)id_codes[p] >= (uint64_t)n_names) {
            int64_t cur = bad.load();
            while (p < cur && !bad.compare_exchange_weak(cur, p)) {
            }
            return;
          }
      });
      const int64_t p = bad.load();
      if (p != INT64_MAX) return ffx_fail(FFX_BAD_ARG, "row %lld holds id code %d, outside the %lld names", (long long)p, id_codes[p], (long long)n_names);
    }
    clock.lap("check_codes");
    const std::string_view name(tag);
    auto format = [&](int64_t q, int64_t r_lo, int64_t r_hi, Out& out) {
      const std::string_view qid = line_of(q_blob, q_start, q);
      const int64_t p_end = pair_off[q] + r_hi;
      for (int64_t r = r_lo; r < r_hi; ++r) {
        const int64_t p = pair_off[q] + r;
        // a row's name is two dependent cache misses in a table of millions (where it starts, its bytes): both are asked for
        // a few rows ahead
        if (p + 16 < p_end) __builtin_prefetch(&name_start[(size_t)id_codes[p + 16]]);
        if (p + 8 < p_end) __builtin_prefetch(name_blob + name_start[(size_t)id_codes[p + 8]]);
        write_line(out, qid, line_of(name_blob, name_start, id_codes[p]), r + 1, scores[p], name);
      }
    };
    return write_lines(path, n_queries, pair_off, format, &clock);
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
}

}  // extern "C"
