// Sanitizer driver for csrc/host_io.cpp (string-id tables + TREC run parser/writer), the only part
// of libffx that parses untrusted input.  Built by `make -C fast-forward-indexes_amd/csrc host_asan`
// with g++ -fsanitize=address,undefined and run by tests/test_host_io.py: a well-formed run, a set
// of hand-written adversarial files, and a few thousand seeded mutations of both (truncations, NUL
// bytes, huge fields, missing columns, non-numeric scores) go through every entry point.  Any
// out-of-bounds access, use-after-free, overflow or misaligned access aborts the process.
#include <algorithm>
#include <cinttypes>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ffx.h"
#include "../../include/ffx_private.h"

// host_io.cpp reports errors through ffx.hip's ffx_fail; the sanitizer build has no HIP side
static std::string g_err;
int ffx_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}

static void write_file(const std::string& path, const std::string& data) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) { perror(path.c_str()); exit(2); }
  fwrite(data.data(), 1, data.size(), f);
  fclose(f);
}

static long g_opened = 0, g_rejected = 0;

// every entry point on one file, with and without an id table
static void exercise(const std::string& path, const std::string& out_path, ffx_ids* ids) {
  for (int pass = 0; pass < 2; ++pass) {
    ffx_run* run = nullptr;
    int64_t bad = -1;
    int rc = ffx_run_open(path.c_str(), pass ? ids : nullptr, &run, &bad);
    if (rc != FFX_OK) {
      ++g_rejected;
      if (run) { fprintf(stderr, "a failed open returned a handle\n"); exit(1); }
      continue;
    }
    ++g_opened;
    int64_t nq = 0, np = 0;
    ffx_run_counts(run, &nq, &np);
    int32_t a = 0, b = 0, c = 0;
    ffx_run_flags(run, &a, &b, &c);
    const char* s = nullptr;
    int32_t len = 0;
    for (int64_t q = -1; q <= nq; ++q)
      if (ffx_run_query_id(run, q, &s, &len) == FFX_OK && len > 0) (void)s[len - 1];
    if (ffx_run_tag(run, &s, &len) == FFX_OK && len > 0) (void)s[len - 1];
    if (!pass) {
      int64_t need = 0;
      ffx_run_ids_blob(run, nullptr, 0, &need);
      std::vector<char> blob((size_t)need + 1);
      ffx_run_ids_blob(run, blob.data(), need, &need);
      ffx_run_ids_blob(run, blob.data(), need > 0 ? need - 1 : 0, &need);  // too small a buffer
    }
    std::vector<int64_t> off((size_t)nq + 1);
    std::vector<int32_t> pid((size_t)np + 1);
    std::vector<float> sp((size_t)np + 1);
    int64_t n = 0;
    const int64_t cutoff = (int64_t)(rnd() % 4);  // 0 = keep all
    ffx_run_export(run, cutoff, off.data(), pass ? pid.data() : nullptr, sp.data(), &n);
    ffx_run_counts(run, &nq, &np);
    if (pass) {
      std::vector<int32_t> order((size_t)np);
      for (int64_t q = 0; q < nq; ++q)
        for (int64_t p = off[(size_t)q]; p < off[(size_t)q + 1]; ++p) order[(size_t)p] = (int32_t)(off[(size_t)q + 1] - 1 - (p - off[(size_t)q]));
      ffx_run_write(run, out_path.c_str(), sp.data(), order.data(), (int64_t)(rnd() % 3), rnd() % 2 ? "tag" : nullptr);
    }
    ffx_run_close(run);
  }
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : "/tmp";
  const int rounds = argc > 2 ? atoi(argv[2]) : 3000;
  const std::string in = dir + "/ffx_fuzz_in.tsv", out = dir + "/ffx_fuzz_out.tsv";

  // ---- id table: fixed-width and packed look-ups, growth across rehashes, odd bytes
  ffx_ids* ids = nullptr;
  if (ffx_ids_create(&ids) != FFX_OK) return 2;
  {
    const int width = 12, n = 5000;
    std::vector<char> buf((size_t)width * n, 0);
    for (int i = 0; i < n; ++i) snprintf(&buf[(size_t)i * width], width, "D%d", i % 4000);  // repeats intern to the same ordinal
    std::vector<int32_t> ord((size_t)n);
    if (ffx_ids_intern(ids, buf.data(), width, n, ord.data()) != FFX_OK) return 2;
    int64_t cnt = 0;
    ffx_ids_count(ids, &cnt);
    if (cnt != 4000 || ord[4500] != 500) { fprintf(stderr, "intern: %" PRId64 " ids, ord %d\n", cnt, ord[4500]); return 1; }
    int64_t miss = -1;
    std::vector<char> q((size_t)width * 3, 0);
    snprintf(&q[0], width, "D17");
    snprintf(&q[(size_t)width], width, "nope");
    memset(&q[(size_t)2 * width], 'x', (size_t)width);  // a field with no NUL terminator at all
    int rc = ffx_ids_lookup(ids, q.data(), width, 3, ord.data(), &miss);
    if (rc != FFX_NOT_FOUND || miss != 1 || ord[0] != 17 || ord[2] != -1) { fprintf(stderr, "lookup rc %d miss %" PRId64 "\n", rc, miss); return 1; }
    const char packed[] = "D1\nD3999\nD4000\n\nD2";
    rc = ffx_ids_lookup_packed(ids, packed, (int64_t)sizeof packed - 1, '\n', 5, ord.data(), &miss);
    if (rc != FFX_NOT_FOUND || miss != 2 || ord[1] != 3999 || ord[4] != 2) { fprintf(stderr, "packed rc %d miss %" PRId64 "\n", rc, miss); return 1; }
    if (ffx_ids_lookup_packed(ids, packed, (int64_t)sizeof packed - 1, '\n', 4, ord.data(), &miss) != FFX_BAD_ARG) return 1;  // wrong count
    if (ffx_ids_lookup_packed(ids, packed, (int64_t)sizeof packed - 1, '\n', 9, ord.data(), &miss) != FFX_BAD_ARG) return 1;
    const char* s = nullptr;
    int32_t len = 0;
    if (ffx_ids_get(ids, 3999, &s, &len) != FFX_OK || len != 5 || memcmp(s, "D3999", 5) != 0) return 1;
    if (ffx_ids_get(ids, 4000, &s, &len) == FFX_OK || ffx_ids_get(ids, -1, &s, &len) == FFX_OK) return 1;
  }

  // ---- well-formed and hand-written adversarial runs
  std::string good;
  for (int q = 0; q < 7; ++q)
    for (int r = 0; r < 9; ++r) {
      char line[128];
      snprintf(line, sizeof line, "q%d\tQ0\tD%d\t%d\t%g\trun-x\n", q, (q * 37 + r * 11) % 4000, r + 1, 20.0 - r * 1.25 + q * 0.01);
      good += line;
    }
  std::vector<std::string> seeds = {
      good,
      "",                                       // empty file
      "\n\n\n",                                 // blank lines only
      "q1 Q0 D1 1 0.5 t",                       // no trailing newline
      "q1 Q0 D1 1 0.5\n",                       // five fields
      "q1 Q0 D1 1 0.5 t extra more\n",          // more than six
      "q1 Q0 D1 1\n",                           // too few
      "q1 Q0 D1 1 abc t\n",                     // score is text
      "q1 Q0 D1 1 0x1A t\nq1 Q0 D2 2 0x1p3 t\n",  // hex floats: strtod would take them, pandas does not
      "q1 Q0 D1 1 nan(123) t\nq1 Q0 D2 2 NAN t\n",
      "q1 Q0 D1 1 nan t\nq1 Q0 D2 2 NA t\nq2 Q0 D3 1 1.5 t\n",  // pandas NA words: rows dropped
      "q1 Q0 D1 1 inf t\nq1 Q0 D2 2 -Infinity t\nq1 Q0 D3 3 1e400 t\nq1 Q0 D4 4 1e-400 t\n",
      "q1 Q0 D99999 1 0.5 t\n",                 // id not in the table
      std::string("q1 Q0 D1 1 0.5 t\n") + std::string(1 << 16, 'A') + " Q0 D2 1 0.25 t\n",  // a 64 KiB query id
      "q1 Q0 " + std::string(1 << 16, 'B') + " 1 0.25 t\n",
      std::string("q1 Q0 D1 1 0.5 t\0q2 Q0 D2 1 0.5 t\n", 34),  // NUL inside the file
      "q1\tQ0\tD1\t1\t\"0.5\"\tt\n",           // quotes
      "  q1   Q0   D1   1   0.5   t  \r\n",     // CRLF and runs of blanks
      "\xff\xfe\xfd Q0 D1 1 0.5 t\n",           // not UTF-8
      "q1 Q0 D1 1 0.5 t\nq2 Q0 D1 1 0.7 t\nq1 Q0 D2 2 0.1 t\n",  // a query that comes back later
  };
  for (const std::string& s : seeds) {
    write_file(in, s);
    exercise(in, out, ids);
  }
  if (ffx_run_open((dir + "/does-not-exist.tsv").c_str(), nullptr, nullptr, nullptr) == FFX_OK) return 1;

  // ---- seeded mutations
  for (int it = 0; it < rounds; ++it) {
    std::string s = seeds[rnd() % seeds.size()];
    const int edits = 1 + (int)(rnd() % 4);
    for (int e = 0; e < edits && !s.empty(); ++e) {
      const size_t at = (size_t)(rnd() % s.size());
      switch (rnd() % 7) {
        case 0: s.resize(at); break;                                        // truncate
        case 1: s[at] = (char)(rnd() & 0xFF); break;                        // flip a byte
        case 2: s[at] = '\0'; break;
        case 3: s.insert(at, std::string((size_t)(rnd() % 300), (char)('0' + rnd() % 75))); break;
        case 4: s.erase(at, (size_t)(rnd() % 40)); break;
        case 5: s[at] = "\t\n \r"[rnd() % 4]; break;
        default: s.insert(at, s.substr((size_t)(rnd() % s.size()), (size_t)(rnd() % 200))); break;  // duplicate a piece
      }
    }
    write_file(in, s);
    exercise(in, out, ids);
  }

  // ---- Ranking.save without a parsed run
  {
    const char qb[] = "qa\nqb", ib[] = "D1\nD2\nD3";
    const int64_t off[] = {0, 2, 3};
    const float sc[] = {1.5f, 0.25f, 1e-7f};
    if (ffx_run_save(out.c_str(), qb, 5, 2, off, ib, 8, 3, sc, "name") != FFX_OK) { fprintf(stderr, "save: %s\n", g_err.c_str()); return 1; }
    const int64_t bad_off[] = {0, 5, 3};
    (void)ffx_run_save(out.c_str(), qb, 5, 2, bad_off, ib, 8, 3, sc, "name");   // must be refused, not read past the scores
    const int64_t off3[] = {0, 2, 3, 3};
    if (ffx_run_save(out.c_str(), qb, 5, 3, off3, ib, 8, 3, sc, "name") == FFX_OK) return 1;  // fewer query ids than queries
    if (ffx_run_save(out.c_str(), qb, 5, 2, bad_off, ib, 8, 3, sc, "name") == FFX_OK) return 1;
  }
  // ---- the coded writer, the string coder and the threaded gathers (round 2)
  {
    const char qb[] = "qa\nqb", names[] = "D1\nD2\nD3";
    const int64_t off[] = {0, 2, 3};
    const float sc[] = {1.5f, 0.25f, 1e-7f};
    const int32_t codes[] = {2, 0, 1}, bad_codes[] = {2, 3, 1}, neg_codes[] = {2, -1, 1};
    if (ffx_run_save_coded(out.c_str(), qb, 5, 2, off, names, 8, 3, codes, 3, sc, "name") != FFX_OK) return 1;
    if (ffx_run_save_coded(out.c_str(), qb, 5, 2, off, names, 8, 3, bad_codes, 3, sc, "name") == FFX_OK) return 1;
    if (ffx_run_save_coded(out.c_str(), qb, 5, 2, off, names, 8, 3, neg_codes, 3, sc, "name") == FFX_OK) return 1;
    if (ffx_run_save_coded(out.c_str(), qb, 5, 2, off, names, 8, 4, codes, 3, sc, "name") == FFX_OK) return 1;  // fewer names than announced
    for (int round = 0; round < 40; ++round) {
      const int64_t n = 1 + (int64_t)(rnd() % 5000);
      std::string blob;
      std::vector<std::string> items;
      for (int64_t i = 0; i < n; ++i) {
        std::string w = rnd() % 9 == 0 ? std::string() : "k" + std::to_string(rnd() % (1 + n / 3));
        if (rnd() % 50 == 0) w += std::string(1 + rnd() % 40, 'x');
        items.push_back(w);
        if (i) blob.push_back('\n');
        blob += w;
      }
      std::vector<int32_t> code((size_t)n);
      char* got = nullptr;
      int64_t got_bytes = 0, got_n = 0;
      const int threads = 1 + (int)(rnd() % 5);
      if (ffx_strings_code(blob.data(), (int64_t)blob.size(), '\n', n, threads, code.data(), &got, &got_bytes, &got_n) != FFX_OK) return 1;
      std::vector<std::string> distinct;
      {
        std::string cur;
        for (int64_t b = 0; b <= got_bytes; ++b) {
          if (b == got_bytes || got[b] == '\n') {
            distinct.push_back(cur);
            cur.clear();
          } else {
            cur.push_back(got[b]);
          }
        }
      }
      if (got_n > 0 && (int64_t)distinct.size() != got_n) return 1;
      for (int64_t i = 0; i < n; ++i)
        if (code[(size_t)i] < 0 || code[(size_t)i] >= got_n || distinct[(size_t)code[(size_t)i]] != items[(size_t)i]) return 1;
      // the same strings back to back with their offsets, in ANOTHER row order: the numbering depends on the set of
      // strings only, so string for string the numbers must agree with the joined-buffer call above
      {
        std::vector<int64_t> order((size_t)n), offs((size_t)n + 1, 0);
        for (int64_t i = 0; i < n; ++i) order[(size_t)i] = i;
        for (int64_t i = n - 1; i > 0; --i) std::swap(order[(size_t)i], order[(size_t)(rnd() % (uint64_t)(i + 1))]);
        std::string packed;
        for (int64_t i = 0; i < n; ++i) {
          packed += items[(size_t)order[(size_t)i]];
          offs[(size_t)i + 1] = (int64_t)packed.size();
        }
        std::vector<int32_t> code2((size_t)n);
        char* got2 = nullptr;
        int64_t bytes2 = 0, n2 = 0;
        if (ffx_strings_code_offsets(packed.data(), offs.data(), n, '\n', 1 + (int)(rnd() % 5), code2.data(), &got2, &bytes2, &n2) != FFX_OK) return 1;
        if (n2 != got_n || bytes2 != got_bytes || (got_bytes > 0 && memcmp(got, got2, (size_t)got_bytes) != 0)) return 1;
        for (int64_t i = 0; i < n; ++i)
          if (code2[(size_t)i] != code[(size_t)order[(size_t)i]]) return 1;
        ffx_buffer_free(got2);
        offs[(size_t)(n / 2) + 1] = offs[(size_t)(n / 2)] - 1;  // not monotone: refused
        if (n > 2 && offs[(size_t)(n / 2)] > 0 && ffx_strings_code_offsets(packed.data(), offs.data(), n, '\n', 2, code2.data(), &got2, &bytes2, &n2) == FFX_OK) return 1;
      }
      ffx_buffer_free(got);
      if (ffx_strings_code(blob.data(), (int64_t)blob.size(), '\n', n + 1, threads, code.data(), &got, &got_bytes, &got_n) == FFX_OK) return 1;  // wrong count
      // numbering 8-byte values (object addresses of a frame's column): equal values share a number, first_row names a holder
      {
        std::vector<uint64_t> vals((size_t)n);
        for (auto& v : vals) v = 0x7f0000000000ull + 48 * (rnd() % (uint64_t)(1 + n / 4));
        std::vector<int32_t> vc((size_t)n);
        std::vector<int64_t> first((size_t)n);
        int64_t nd = 0;
        if (ffx_host_code_u64(vals.data(), n, 1 + (int)(rnd() % 5), vc.data(), first.data(), &nd) != FFX_OK || nd < 1 || nd > n) return 1;
        for (int64_t i = 0; i < n; ++i)
          if (vc[(size_t)i] < 0 || vc[(size_t)i] >= nd || vals[(size_t)first[(size_t)vc[(size_t)i]]] != vals[(size_t)i]) return 1;
        for (int64_t c = 0; c < nd; ++c)
          if (vc[(size_t)first[(size_t)c]] != c || (c > 0 && vals[(size_t)first[(size_t)c]] <= vals[(size_t)first[(size_t)c - 1]])) return 1;  // ascending with the values
      }
      // every query ordered and the queries' blocks laid out in reverse in one pass
      {
        const int64_t nq = 1 + (int64_t)(rnd() % 7);
        std::vector<int64_t> off((size_t)nq + 1, 0), dst((size_t)nq);
        for (int64_t q = 0; q < nq; ++q) off[(size_t)q + 1] = off[(size_t)q] + (q + 1 == nq ? n - off[(size_t)q] : (int64_t)(rnd() % (uint64_t)(1 + (n - off[(size_t)q]) / 2)));
        int64_t at = 0;
        for (int64_t q = nq - 1; q >= 0; --q) {
          dst[(size_t)q] = at;
          at += off[(size_t)q + 1] - off[(size_t)q];
        }
        std::vector<float> sc((size_t)n);
        for (auto& v : sc) v = (float)(rnd() % 50) * 0.5f;
        std::vector<int32_t> plain((size_t)n), moved((size_t)n);
        if (ffx_host_order(sc.data(), off.data(), nq, 1 + (int)(rnd() % 4), plain.data()) != FFX_OK) return 1;
        if (ffx_host_order_blocks(sc.data(), off.data(), dst.data(), nq, 1 + (int)(rnd() % 4), moved.data()) != FFX_OK) return 1;
        for (int64_t q = 0; q < nq; ++q)
          for (int64_t i = 0; i < off[(size_t)q + 1] - off[(size_t)q]; ++i)
            if (moved[(size_t)(dst[(size_t)q] + i)] != plain[(size_t)(off[(size_t)q] + i)]) return 1;
        dst[0] = n;  // outside the output: refused
        if (off[1] > 0 && ffx_host_order_blocks(sc.data(), off.data(), dst.data(), nq, 2, moved.data()) == FFX_OK) return 1;
      }
      // gathers and the join-free interpolate on the same sizes
      std::vector<float> src((size_t)n), dense((size_t)n), res((size_t)n);
      std::vector<int32_t> perm((size_t)n);
      for (int64_t i = 0; i < n; ++i) src[(size_t)i] = (float)(rnd() % 1000) * 0.25f, dense[(size_t)i] = (float)(rnd() % 77), perm[(size_t)i] = (int32_t)i;
      for (int64_t i = n - 1; i > 0; --i) std::swap(perm[(size_t)i], perm[(size_t)(rnd() % (uint64_t)(i + 1))]);
      if (ffx_host_take32(src.data(), n, perm.data(), n, threads, res.data()) != FFX_OK) return 1;
      for (int64_t i = 0; i < n; ++i)
        if (res[(size_t)i] != src[(size_t)perm[(size_t)i]]) return 1;
      int32_t nan = 0;
      if (ffx_host_interpolate(src.data(), dense.data(), perm.data(), n, 0.25f, 0.75f, threads, res.data(), &nan) != FFX_OK || nan) return 1;
      if (n > 1) {
        std::vector<int32_t> twice = perm;
        twice[0] = twice[1];
        if (ffx_host_interpolate(src.data(), dense.data(), twice.data(), n, 0.25f, 0.75f, threads, res.data(), &nan) == FFX_OK) return 1;
        twice[0] = (int32_t)n;
        if (ffx_host_interpolate(src.data(), dense.data(), twice.data(), n, 0.25f, 0.75f, threads, res.data(), &nan) == FFX_OK) return 1;
        if (ffx_host_take32(src.data(), n, twice.data(), n, threads, res.data()) == FFX_OK) return 1;
      }
    }
  }
  // ---- the threaded outer join
  for (int round = 0; round < 30; ++round) {
    const int64_t nq = (int64_t)(rnd() % 40);
    std::vector<int64_t> oa{0}, ob{0};
    std::vector<int32_t> ca, cb;
    for (int64_t g = 0; g < nq; ++g) {
      const int pool = 1 + (int)(rnd() % 300);
      std::vector<int32_t> all((size_t)pool);
      for (int i = 0; i < pool; ++i) all[(size_t)i] = i * 3 + (int)(rnd() % 3);
      for (int side = 0; side < 2; ++side) {
        std::vector<int32_t>& dst = side ? cb : ca;
        std::vector<int32_t> pick;
        for (int32_t c : all)
          if (rnd() % 3) pick.push_back(c);
        for (size_t i = pick.size(); i > 1; --i) std::swap(pick[i - 1], pick[(size_t)(rnd() % i)]);
        if (rnd() % 7 == 0) pick.clear();
        dst.insert(dst.end(), pick.begin(), pick.end());
        (side ? ob : oa).push_back((int64_t)dst.size());
      }
    }
    const size_t room = ca.size() + cb.size();
    std::vector<int64_t> uoff((size_t)nq + 1);
    std::vector<int32_t> uc(room + 1), ua(room + 1), ub(room + 1);
    if (ffx_host_outer_join(oa.data(), ca.data(), ob.data(), cb.data(), nq, 1 + (int)(rnd() % 4), uoff.data(), uc.data(), ua.data(), ub.data()) != FFX_OK) return 1;
    for (int64_t g = 0; g < nq; ++g) {
      int64_t prev = -1;
      int64_t seen_a = 0, seen_b = 0;
      for (int64_t i = uoff[(size_t)g]; i < uoff[(size_t)g + 1]; ++i) {
        if (uc[(size_t)i] <= prev) return 1;  // ascending, no duplicate
        prev = uc[(size_t)i];
        if (ua[(size_t)i] < 0 && ub[(size_t)i] < 0) return 1;
        if (ua[(size_t)i] >= 0) {
          if (ua[(size_t)i] < oa[(size_t)g] || ua[(size_t)i] >= oa[(size_t)g + 1] || ca[(size_t)ua[(size_t)i]] != uc[(size_t)i]) return 1;
          ++seen_a;
        }
        if (ub[(size_t)i] >= 0) {
          if (ub[(size_t)i] < ob[(size_t)g] || ub[(size_t)i] >= ob[(size_t)g + 1] || cb[(size_t)ub[(size_t)i]] != uc[(size_t)i]) return 1;
          ++seen_b;
        }
      }
      if (seen_a != oa[(size_t)g + 1] - oa[(size_t)g] || seen_b != ob[(size_t)g + 1] - ob[(size_t)g]) return 1;
    }
    if (!ca.empty() && oa.size() > 1) {  // an id twice on one side is refused
      for (int64_t g = 0; g < nq; ++g)
        if (oa[(size_t)g + 1] - oa[(size_t)g] >= 2) {
          std::vector<int32_t> twice = ca;
          twice[(size_t)oa[(size_t)g]] = twice[(size_t)oa[(size_t)g] + 1];
          if (ffx_host_outer_join(oa.data(), twice.data(), ob.data(), cb.data(), nq, 2, uoff.data(), uc.data(), ua.data(), ub.data()) == FFX_OK) return 1;
          break;
        }
    }
  }
  // ---- distinct rows of a streamed batch
  for (int round = 0; round < 30; ++round) {
    const int64_t n = 1 + (int64_t)(rnd() % 4000), n_index = 1 + (int64_t)(rnd() % 3000);
    std::vector<int64_t> rows((size_t)n), uniq((size_t)std::min(n, n_index));
    for (auto& r : rows) r = (int64_t)(rnd() % (uint64_t)n_index);
    std::vector<int32_t> inv((size_t)n);
    int64_t count = 0;
    if (ffx_host_unique_rows(rows.data(), n, n_index, 1 + (int)(rnd() % 4), uniq.data(), &count, inv.data()) != FFX_OK) return 1;
    for (int64_t i = 1; i < count; ++i)
      if (uniq[(size_t)i] <= uniq[(size_t)i - 1]) return 1;
    for (int64_t i = 0; i < n; ++i)
      if (inv[(size_t)i] < 0 || inv[(size_t)i] >= count || uniq[(size_t)inv[(size_t)i]] != rows[(size_t)i]) return 1;
    rows[0] = n_index;  // outside the index: refused
    if (ffx_host_unique_rows(rows.data(), n, n_index, 2, uniq.data(), &count, inv.data()) == FFX_OK) return 1;
  }
  // ---- every threaded native once at a size where several threads really run (the pools take a thread per 64 Ki
  // elements): the answer with 4 threads equals the answer with 1.  Under -fsanitize=thread this is the data-race check.
  {
    const int64_t n = 400000, nq = 900;
    std::vector<int64_t> off((size_t)nq + 1, 0);
    for (int64_t q = 1; q <= nq; ++q) off[(size_t)q] = q == nq ? n : std::min<int64_t>(n, off[(size_t)q - 1] + (int64_t)(rnd() % 900));
    std::vector<float> sc((size_t)n), dense((size_t)n);
    std::vector<int32_t> perm((size_t)n);
    for (int64_t i = 0; i < n; ++i) sc[(size_t)i] = (float)(rnd() % 4096) * 0.125f, dense[(size_t)i] = (float)(rnd() % 977), perm[(size_t)i] = (int32_t)i;
    for (int64_t i = n - 1; i > 0; --i) std::swap(perm[(size_t)i], perm[(size_t)(rnd() % (uint64_t)(i + 1))]);
    std::vector<int32_t> o1((size_t)n), o4((size_t)n);
    if (ffx_host_order(sc.data(), off.data(), nq, 1, o1.data()) != FFX_OK || ffx_host_order(sc.data(), off.data(), nq, 4, o4.data()) != FFX_OK || o1 != o4) return 1;
    std::vector<int64_t> dst((size_t)nq);
    for (int64_t q = 0; q < nq; ++q) dst[(size_t)q] = n - off[(size_t)q + 1];  // the queries in reverse
    if (ffx_host_order_blocks(sc.data(), off.data(), dst.data(), nq, 1, o1.data()) != FFX_OK || ffx_host_order_blocks(sc.data(), off.data(), dst.data(), nq, 4, o4.data()) != FFX_OK || o1 != o4) return 1;
    std::vector<float> t1((size_t)n), t4((size_t)n);
    if (ffx_host_take32(sc.data(), n, perm.data(), n, 1, t1.data()) != FFX_OK || ffx_host_take32(sc.data(), n, perm.data(), n, 4, t4.data()) != FFX_OK || t1 != t4) return 1;
    int32_t nan = 0;
    if (ffx_host_interpolate(sc.data(), dense.data(), perm.data(), n, 0.25f, 0.75f, 1, t1.data(), &nan) != FFX_OK || ffx_host_interpolate(sc.data(), dense.data(), perm.data(), n, 0.25f, 0.75f, 4, t4.data(), &nan) != FFX_OK || t1 != t4) return 1;
    // distinct rows
    const int64_t n_index = 250000;
    std::vector<int64_t> rows((size_t)n), u1((size_t)n_index), u4((size_t)n_index);
    for (auto& r : rows) r = (int64_t)(rnd() % (uint64_t)n_index);
    int64_t c1 = 0, c4 = 0;
    if (ffx_host_unique_rows(rows.data(), n, n_index, 1, u1.data(), &c1, o1.data()) != FFX_OK || ffx_host_unique_rows(rows.data(), n, n_index, 4, u4.data(), &c4, o4.data()) != FFX_OK) return 1;
    if (c1 != c4 || o1 != o4 || !std::equal(u1.begin(), u1.begin() + c1, u4.begin())) return 1;
    // ids -> rows -> distinct rows in one pass: equals the two steps done apart, whatever the thread count
    {
      const int64_t n1 = 180000, n2 = 200000;
      std::vector<int32_t> t1((size_t)n1), t2((size_t)n2), ids((size_t)n), two((size_t)n);
      for (auto& v : t1) v = (int32_t)(rnd() % (uint64_t)n2);
      for (auto& v : t2) v = (int32_t)(rnd() % (uint64_t)n_index);
      for (auto& v : ids) v = (int32_t)(rnd() % (uint64_t)n1);
      std::vector<int64_t> apart((size_t)n), ua((size_t)n_index), ub((size_t)n_index);
      for (int64_t i = 0; i < n; ++i) apart[(size_t)i] = t2[(size_t)t1[(size_t)ids[(size_t)i]]];
      int64_t ca = 0, cb = 0;
      if (ffx_host_unique_rows(apart.data(), n, n_index, 1, ua.data(), &ca, o1.data()) != FFX_OK) return 1;
      for (int threads : {1, 4}) {
        if (ffx_host_rows_unique(ids.data(), n, t1.data(), n1, t2.data(), n2, n_index, threads, ub.data(), &cb, two.data()) != FFX_OK) return 1;
        if (ca != cb || o1 != two || !std::equal(ua.begin(), ua.begin() + ca, ub.begin())) return 1;
      }
      for (int64_t i = 0; i < n; ++i) apart[(size_t)i] = t1[(size_t)ids[(size_t)i]];  // one table only
      if (ffx_host_unique_rows(apart.data(), n, n2, 1, ua.data(), &ca, o1.data()) != FFX_OK) return 1;
      if (ffx_host_rows_unique(ids.data(), n, t1.data(), n1, nullptr, 0, n2, 4, ub.data(), &cb, two.data()) != FFX_OK || ca != cb || o1 != two) return 1;
      ids[7] = (int32_t)n1;  // an id outside the table, an entry outside the second table, a row outside the index: refused
      if (ffx_host_rows_unique(ids.data(), n, t1.data(), n1, t2.data(), n2, n_index, 4, ub.data(), &cb, two.data()) == FFX_OK) return 1;
      ids[7] = 0;
      const int32_t keep = t1[0];
      t1[0] = (int32_t)n2;
      if (ffx_host_rows_unique(ids.data(), n, t1.data(), n1, t2.data(), n2, n_index, 4, ub.data(), &cb, two.data()) == FFX_OK) return 1;
      t1[0] = keep;
      if (ffx_host_rows_unique(ids.data(), n, t1.data(), n1, t2.data(), n2, n_index / 2, 4, ub.data(), &cb, two.data()) == FFX_OK) return 1;
    }
    // all rows of every candidate's document: offsets, then rows -> distinct rows + slots, against the steps done apart
    {
      const int64_t nd = 60000;
      std::vector<int32_t> eoff((size_t)nd + 1, 0), ids((size_t)n);
      for (int64_t d = 0; d < nd; ++d) eoff[(size_t)d + 1] = eoff[(size_t)d] + 1 + (int32_t)(rnd() % 5);
      const int64_t ne = eoff[(size_t)nd];
      std::vector<int32_t> erow((size_t)ne);
      for (auto& v : erow) v = (int32_t)(rnd() % (uint64_t)n_index);
      for (auto& v : ids) v = (int32_t)(rnd() % (uint64_t)nd);
      std::vector<int64_t> so1((size_t)n + 1), so4((size_t)n + 1);
      if (ffx_host_segment_offsets(ids.data(), n, eoff.data(), nd, 1, so1.data()) != FFX_OK || ffx_host_segment_offsets(ids.data(), n, eoff.data(), nd, 4, so4.data()) != FFX_OK || so1 != so4) return 1;
      const int64_t R2 = so1[(size_t)n];
      std::vector<int64_t> apart((size_t)R2), ua((size_t)n_index), ub((size_t)n_index);
      for (int64_t i = 0; i < n; ++i)
        for (int64_t k = so1[(size_t)i]; k < so1[(size_t)i + 1]; ++k) apart[(size_t)k] = erow[(size_t)(eoff[(size_t)ids[(size_t)i]] + (k - so1[(size_t)i]))];
      std::vector<int32_t> ia((size_t)R2), ib((size_t)R2);
      int64_t ca = 0, cb = 0;
      if (ffx_host_unique_rows(apart.data(), R2, n_index, 1, ua.data(), &ca, ia.data()) != FFX_OK) return 1;
      for (int threads : {1, 4}) {
        if (ffx_host_segment_rows_unique(ids.data(), n, eoff.data(), nd, erow.data(), ne, so1.data(), n_index, threads, ub.data(), &cb, ib.data()) != FFX_OK) return 1;
        if (ca != cb || ia != ib || !std::equal(ua.begin(), ua.begin() + ca, ub.begin())) return 1;
      }
      ids[3] = (int32_t)nd;  // a document outside the table; offsets that do not belong to the ids; a row outside the index: refused
      if (ffx_host_segment_offsets(ids.data(), n, eoff.data(), nd, 4, so4.data()) == FFX_OK) return 1;
      if (ffx_host_segment_rows_unique(ids.data(), n, eoff.data(), nd, erow.data(), ne, so1.data(), n_index, 4, ub.data(), &cb, ib.data()) == FFX_OK) return 1;
      ids[3] = 0;
      if (ffx_host_segment_offsets(ids.data(), n, eoff.data(), nd, 4, so4.data()) != FFX_OK) return 1;
      std::vector<int32_t> ic((size_t)so4[(size_t)n]);
      if (ffx_host_segment_rows_unique(ids.data(), n, eoff.data(), nd, erow.data(), ne, so4.data(), n_index / 2, 4, ub.data(), &cb, ic.data()) == FFX_OK) return 1;
    }
    // look-ups of a packed buffer large enough for several threads (byte ranges, strings numbered by the separators in
    // front of them): every ordinal right, empty strings and strings nobody interned reported as -1, the first one named
    {
      ffx_ids* big = nullptr;
      if (ffx_ids_create(&big) != FFX_OK) return 1;
      const int64_t n_ids = 150000, n_q = 400000;
      std::string all;
      for (int64_t i = 0; i < n_ids; ++i) {
        if (i) all.push_back('\n');
        all += "doc" + std::to_string(i * 7);
      }
      std::vector<int32_t> got((size_t)n_ids);
      if (ffx_ids_intern_packed(big, all.data(), (int64_t)all.size(), '\n', n_ids, got.data()) != FFX_OK) return 1;
      std::string asked;
      std::vector<int32_t> want((size_t)n_q), found((size_t)n_q);
      int64_t first_bad = -1;
      for (int64_t i = 0; i < n_q; ++i) {
        if (i) asked.push_back('\n');
        const uint64_t r = rnd();
        if (r % 1000 == 0 && i > 5000) {
          want[(size_t)i] = -1;
          if (first_bad < 0) first_bad = i;
          if (r % 2000 == 0) asked += "nobody" + std::to_string(i);  // (else: an empty string)
        } else {
          const int64_t o = (int64_t)(r % (uint64_t)n_ids);
          want[(size_t)i] = (int32_t)o;
          asked += "doc" + std::to_string(o * 7);
        }
      }
      int64_t miss = -1;
      const int rc = ffx_ids_lookup_packed(big, asked.data(), (int64_t)asked.size(), '\n', n_q, found.data(), &miss);
      if (rc != FFX_NOT_FOUND || miss != first_bad || found != want) return 1;
      if (ffx_ids_lookup_packed(big, asked.data(), (int64_t)asked.size(), '\n', n_q - 1, found.data(), &miss) != FFX_BAD_ARG) return 1;
      asked.push_back('\n');  // one more (empty) string at the very end
      found.push_back(0);
      if (ffx_ids_lookup_packed(big, asked.data(), (int64_t)asked.size(), '\n', n_q + 1, found.data(), &miss) != FFX_NOT_FOUND || found[(size_t)n_q] != -1) return 1;
      ffx_ids_destroy(big);
    }
    // 8-byte values (pointers of an object column) numbered in ascending order
    std::vector<uint64_t> vals((size_t)n);
    for (auto& v : vals) v = 0x7f0000000000ull + 48 * (uint64_t)(rnd() % 50000);
    std::vector<int64_t> f1((size_t)n), f4((size_t)n);
    if (ffx_host_code_u64(vals.data(), n, 1, o1.data(), f1.data(), &c1) != FFX_OK || ffx_host_code_u64(vals.data(), n, 4, o4.data(), f4.data(), &c4) != FFX_OK || c1 != c4 || o1 != o4) return 1;
    for (int64_t c = 0; c < c1; ++c)
      if (vals[(size_t)f4[(size_t)c]] != vals[(size_t)f1[(size_t)c]] || (c > 0 && vals[(size_t)f4[(size_t)c]] <= vals[(size_t)f4[(size_t)c - 1]])) return 1;
    // ... and a column of long RUNS (the run-head shortcut): the numbers the general scheme gives, by 1 and by 4 threads; the
    // per-row run numbers (ffx_host_fill_runs) against a plain loop
    {
      std::vector<int64_t> run_off{0};
      std::vector<uint64_t> runs;
      while (run_off.back() < n) {
        const int64_t len = std::min<int64_t>(n - run_off.back(), 1 + (int64_t)(rnd() % 9000));
        runs.insert(runs.end(), (size_t)len, 0x7f0000000000ull + 48 * (uint64_t)(rnd() % 300));
        run_off.push_back(run_off.back() + len);
      }
      std::vector<int32_t> r1((size_t)n), r4((size_t)n);
      std::vector<int64_t> g1((size_t)n), g4((size_t)n);
      int64_t d1 = 0, d4 = 0;
      if (ffx_host_code_u64(runs.data(), n, 1, r1.data(), g1.data(), &d1) != FFX_OK || ffx_host_code_u64(runs.data(), n, 4, r4.data(), g4.data(), &d4) != FFX_OK ||
          d1 != d4 || r1 != r4)
        return 1;
      for (int64_t i = 0; i < n; ++i)
        if (r4[(size_t)i] < 0 || r4[(size_t)i] >= d4 || runs[(size_t)g4[(size_t)r4[(size_t)i]]] != runs[(size_t)i]) return 1;
      for (int64_t c = 1; c < d4; ++c)
        if (runs[(size_t)g4[(size_t)c]] <= runs[(size_t)g4[(size_t)c - 1]]) return 1;
      const int64_t n_runs = (int64_t)run_off.size() - 1;
      std::vector<int64_t> numbers((size_t)n, -1);
      if (ffx_host_fill_runs(run_off.data(), n_runs, 4, numbers.data()) != FFX_OK) return 1;
      for (int64_t r = 0; r < n_runs; ++r)
        for (int64_t i = run_off[(size_t)r]; i < run_off[(size_t)r + 1]; ++i)
          if (numbers[(size_t)i] != r) return 1;
      run_off[1] = run_off[2] + 1;  // not monotone: refused
      if (n_runs > 2 && ffx_host_fill_runs(run_off.data(), n_runs, 2, numbers.data()) != FFX_BAD_ARG) return 1;
    }
    // strings: joined by a separator and back to back with offsets; the numbering depends on the set of strings only
    std::string blob, packed;
    std::vector<int64_t> soff{0};
    const int64_t ns = 300000;
    for (int64_t i = 0; i < ns; ++i) {
      const std::string w = "doc" + std::to_string(rnd() % 70000);
      if (i) blob.push_back('\n');
      blob += w;
      packed += w;
      soff.push_back((int64_t)packed.size());
    }
    std::vector<int32_t> s1((size_t)ns), s4((size_t)ns), s5((size_t)ns);
    char *b1 = nullptr, *b4 = nullptr, *b5 = nullptr;
    int64_t nb1 = 0, nb4 = 0, nb5 = 0, nn1 = 0, nn4 = 0, nn5 = 0;
    if (ffx_strings_code(blob.data(), (int64_t)blob.size(), '\n', ns, 1, s1.data(), &b1, &nb1, &nn1) != FFX_OK) return 1;
    if (ffx_strings_code(blob.data(), (int64_t)blob.size(), '\n', ns, 4, s4.data(), &b4, &nb4, &nn4) != FFX_OK) return 1;
    if (ffx_strings_code_offsets(packed.data(), soff.data(), ns, '\n', 4, s5.data(), &b5, &nb5, &nn5) != FFX_OK) return 1;
    if (nn1 != nn4 || nn1 != nn5 || nb1 != nb4 || nb1 != nb5 || s1 != s4 || s1 != s5 || memcmp(b1, b4, (size_t)nb1) != 0 || memcmp(b1, b5, (size_t)nb1) != 0) return 1;
    ffx_buffer_free(b1);
    ffx_buffer_free(b4);
    ffx_buffer_free(b5);
    // owners of fixed-width id columns
    const int width = 12;
    std::vector<char> ids_col((size_t)ns * width, 0);
    for (int64_t i = 0; i < ns; ++i) snprintf(&ids_col[(size_t)i * width], width, "D%d", (int)(rnd() % 90000));
    std::vector<uint8_t> w1((size_t)ns), w4((size_t)ns);
    if (ffx_host_owner_crc32(ids_col.data(), width, nullptr, 0, ns, 8, 1, w1.data()) != FFX_OK || ffx_host_owner_crc32(ids_col.data(), width, nullptr, 0, ns, 8, 4, w4.data()) != FFX_OK || w1 != w4) return 1;
    // the outer join over many queries
    std::vector<int64_t> oa{0}, ob{0};
    std::vector<int32_t> ca, cb;
    for (int64_t g = 0; g < 2000; ++g) {
      for (int side = 0; side < 2; ++side) {
        std::vector<int32_t>& d = side ? cb : ca;
        for (int32_t c = 0; c < 300; ++c)
          if (rnd() % 3) d.push_back(c * 7 + (int32_t)g % 5);
        (side ? ob : oa).push_back((int64_t)d.size());
      }
    }
    const size_t room = ca.size() + cb.size() + 1;
    std::vector<int64_t> uo1(2001), uo4(2001);
    std::vector<int32_t> uc1(room), ua1(room), ub1(room), uc4(room), ua4(room), ub4(room);
    if (ffx_host_outer_join(oa.data(), ca.data(), ob.data(), cb.data(), 2000, 1, uo1.data(), uc1.data(), ua1.data(), ub1.data()) != FFX_OK) return 1;
    if (ffx_host_outer_join(oa.data(), ca.data(), ob.data(), cb.data(), 2000, 4, uo4.data(), uc4.data(), ua4.data(), ub4.data()) != FFX_OK) return 1;
    const size_t nu = (size_t)uo1[2000];
    if (uo1 != uo4 || !std::equal(uc1.begin(), uc1.begin() + nu, uc4.begin()) || !std::equal(ua1.begin(), ua1.begin() + nu, ua4.begin()) || !std::equal(ub1.begin(), ub1.begin() + nu, ub4.begin())) return 1;
  }
  ffx_ids_destroy(ids);
  remove(in.c_str());
  remove(out.c_str());
  printf("host_io sanitizer run ok: %ld files parsed, %ld rejected\n", g_opened, g_rejected);
  return g_opened > 0 && g_rejected > 0 ? 0 : 1;
}
