// host_util.h -- what the host-side translation units of libffx (host_io.cpp: id tables, string numbering, array passes;
// run_file.cpp: TREC run parser and writer) share: the thread helper, the hash, the id table, an uninitialised buffer
// and the stage clock.  Plain C++17, no device code; also compiled by g++ for the sanitizer builds.
#ifndef FFX_HOST_UTIL_H_
#define FFX_HOST_UTIL_H_

#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include "../../include/ffx.h"
#include "../../include/ffx_private.h"

int ffx_fail(int code, const char* fmt, ...);  // ffx.hip: records the thread's last error

namespace ffx_host {

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

// threads for a pass over n items of which one thread should get `grain` at least (n_threads <= 0: every usable core)
inline int threads_for(int64_t n, int64_t grain, int32_t n_threads = 0, int cap = 64) {
  const int T = n_threads > 0 ? n_threads : usable_cores();
  return (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)T, (int64_t)cap, n / std::max<int64_t>(1, grain) + 1}));
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

inline uint64_t fnv1a(const char* p, size_t n) {
  uint64_t h = 1469598103934665603ull;
  for (size_t i = 0; i < n; ++i) {
    h ^= (unsigned char)p[i];
    h *= 1099511628211ull;
  }
  return h ^ (h >> 29);
}

// The hash the string numbering (string_code.h) shards and orders by.  Up to eight bytes -- "D1234567", MS MARCO's numeric
// ids -- are taken as one integer and mixed by a BIJECTION of 64-bit words (after adding a multiple of the length): two such
// strings of one length with one hash are the same string, no byte needs comparing.  Longer strings: fnv-1a, mixed the same
// way (fnv's high bits are weak on short numeric strings and the shard is taken from them).
inline uint64_t mix64(uint64_t h) {
  h ^= h >> 30;
  h *= 0xbf58476d1ce4e5b9ull;
  h ^= h >> 27;
  h *= 0x94d049bb133111ebull;
  return h ^ (h >> 31);
}
inline uint64_t name_hash(const char* p, size_t n) {
  if (n <= 8) {
    uint64_t k = 0;
    memcpy(&k, p, n);
    return mix64(k + (uint64_t)n * 0x9E3779B97F4A7C15ull);
  }
  return mix64(fnv1a(p, n));
}

// n elements of T, NOT initialised (a std::vector of tens of millions of elements is zeroed by the one thread that makes
// it, and that thread takes every page fault; here the threads that fill the buffer touch its pages first)
template <class T>
struct Buf {
  T* p = nullptr;
  size_t n = 0;
  Buf() = default;
  explicit Buf(size_t count) { alloc(count); }
  Buf(const Buf&) = delete;
  Buf& operator=(const Buf&) = delete;
  Buf(Buf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr, o.n = 0; }
  Buf& operator=(Buf&& o) noexcept {
    if (this != &o) {
      free(p);
      p = o.p, n = o.n;
      o.p = nullptr, o.n = 0;
    }
    return *this;
  }
  ~Buf() { free(p); }
  void alloc(size_t count) {
    free(p);
    p = nullptr, n = 0;
    if (count == 0) return;
    if (count > SIZE_MAX / sizeof(T)) throw std::bad_alloc();
    p = static_cast<T*>(malloc(count * sizeof(T)));
    if (!p) throw std::bad_alloc();
    n = count;
  }
  void release() {
    free(p);
    p = nullptr, n = 0;
  }
  T* data() { return p; }
  const T* data() const { return p; }
  size_t size() const { return n; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};

// Where the time of one host call went: "stage=seconds;stage=seconds;..." of the calling thread's last instrumented call
// (ffx_host_stage_report), so that a bench line can say what a second of Ranking.from_file is made of.
std::string& stage_report();
struct StageClock {
  std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
  std::string text;
  void lap(const char* name) {
    const auto now = std::chrono::steady_clock::now();
    char b[96];
    snprintf(b, sizeof b, "%s=%.6f;", name, std::chrono::duration<double>(now - last).count());
    text += b;
    last = now;
  }
  ~StageClock() { stage_report() = text; }
};

}  // namespace ffx_host

// insertion-ordered interning table: ordinal = order of first insertion.  Open addressing; a slot holds the ordinal + 1
// (0: empty) under the upper half of the string's hash, so that a probe that meets another string's slot goes on without
// touching that string.
struct ffx_ids {
  std::vector<char> arena;
  std::vector<uint64_t> off;   // off[i]..off[i+1] = bytes of string i
  std::vector<uint64_t> slots;  // (hash >> 32) << 32 | ordinal + 1  (huge pages for these arrays: tried twice on the box's host, no gain)
  size_t mask = 0;

  ffx_ids() { off.push_back(0); rehash(1024); }
  int64_t size() const { return (int64_t)off.size() - 1; }
  std::string_view get(int32_t i) const { return {arena.data() + off[(size_t)i], (size_t)(off[(size_t)i + 1] - off[(size_t)i])}; }
  static uint64_t tag_of(uint64_t h) { return h & 0xFFFFFFFF00000000ull; }
  void rehash(size_t cap) {
    slots.assign(cap, 0);
    mask = cap - 1;
    for (int32_t i = 0; i < (int32_t)size(); ++i) {
      const std::string_view s = get(i);
      const uint64_t h = ffx_host::fnv1a(s.data(), s.size());
      size_t at = h & mask;
      while (slots[at] != 0) at = (at + 1) & mask;
      slots[at] = tag_of(h) | (uint32_t)(i + 1);
    }
  }
  // the ordinal of s given its hash (-1: not there)
  int32_t find_hashed(std::string_view s, uint64_t h) const {
    const uint64_t tag = tag_of(h);
    size_t at = h & mask;
    while (true) {
      const uint64_t v = slots[at];
      if (v == 0) return -1;
      if (tag_of(v) == tag) {
        const int32_t i = (int32_t)(uint32_t)v - 1;
        if (get(i) == s) return i;
      }
      at = (at + 1) & mask;
    }
  }
  int32_t find(std::string_view s) const { return find_hashed(s, ffx_host::fnv1a(s.data(), s.size())); }
  // room for n_more strings of n_bytes in all: the arrays grow once and the slots are laid out once (a table that doubles its
  // way to 8.8 M strings moves every array ~twice and re-inserts every string ~twice)
  void expect(int64_t n_more, int64_t n_bytes) {
    arena.reserve(arena.size() + (size_t)n_bytes);
    off.reserve(off.size() + (size_t)n_more);
    size_t cap = slots.size();
    while (((size_t)size() + (size_t)n_more) * 10 > cap * 6) cap *= 2;
    if (cap != slots.size()) rehash(cap);
  }
  int32_t intern(std::string_view s) {
    const uint64_t h = ffx_host::fnv1a(s.data(), s.size());
    const uint64_t tag = tag_of(h);
    size_t at = h & mask;
    while (true) {
      const uint64_t v = slots[at];
      if (v == 0) break;
      if (tag_of(v) == tag) {
        const int32_t i = (int32_t)(uint32_t)v - 1;
        if (get(i) == s) return i;
      }
      at = (at + 1) & mask;
    }
    const int32_t id = (int32_t)size();
    arena.insert(arena.end(), s.begin(), s.end());
    off.push_back(arena.size());
    slots[at] = tag | (uint32_t)(id + 1);
    if ((size_t)size() * 10 > slots.size() * 6) rehash(slots.size() * 2);
    return id;
  }
};

#endif  // FFX_HOST_UTIL_H_
