#include <emmintrin.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static void nt_copy(void* d, const void* s, size_t n) {
  char* dst = (char*)d; const char* src = (const char*)s;
  size_t head = (16 - ((uintptr_t)dst & 15)) & 15;
  if (head > n) head = n;
  memcpy(dst, src, head); dst += head; src += head; n -= head;
  size_t blocks = n / 64;
  for (size_t i = 0; i < blocks; ++i) {
    __m128i a = _mm_loadu_si128((const __m128i*)src), b = _mm_loadu_si128((const __m128i*)(src + 16));
    __m128i c = _mm_loadu_si128((const __m128i*)(src + 32)), e = _mm_loadu_si128((const __m128i*)(src + 48));
    _mm_stream_si128((__m128i*)dst, a); _mm_stream_si128((__m128i*)(dst + 16), b);
    _mm_stream_si128((__m128i*)(dst + 32), c); _mm_stream_si128((__m128i*)(dst + 48), e);
    src += 64; dst += 64;
  }
  _mm_sfence();
  memcpy(dst, src, n - blocks * 64);
}
int main(int argc, char** argv) {
  int T = argc > 1 ? atoi(argv[1]) : 1;
  size_t total = size_t(1) << 30, piece = 4 << 20;
  char* src = (char*)aligned_alloc(4096, total); char* dst = (char*)aligned_alloc(4096, total);
  memset(src, 1, total); memset(dst, 2, total);
  for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 2; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back([=] {
      for (size_t at = t * piece; at < total; at += T * piece) { if (mode) nt_copy(dst + at, src + at, piece); else memcpy(dst + at, src + at, piece); }
    });
    for (auto& x : th) x.join();
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("T=%d %s: %.1f GB/s\n", T, mode ? "nt_copy" : "memcpy", total / s / 1e9);
  }
}
