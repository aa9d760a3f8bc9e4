// staged_xfer_probe.cpp -- the library's staged transfers (csrc/host_xfer.h) alone, against the runtime's page-locked copies:
// ordinary host memory <-> device, 8 .. 256 MB, best of 5.  Build and run (GPU box):
//   hipcc -O2 -std=c++17 -o /tmp/staged_xfer_probe tools/probes/staged_xfer_probe.cpp \
//         -Lfast-forward-indexes_amd/fast_forward_amd -l:libffx_experiments.so -Wl,-rpath,$PWD/fast-forward-indexes_amd/fast_forward_amd -pthread
//   FFX_DEBUG_VARIANT=xfer_lanes=4,xfer_piece_kb=4096 /tmp/staged_xfer_probe
// (with libffx_experiments.so the knobs xfer_lanes / xfer_piece_kb apply; with libffx.so they are compiled out)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../fast-forward-indexes_amd/csrc/host_xfer.h"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t most = size_t(256) << 20;
  char *dev = nullptr, *pinned = nullptr;
  if (hipSetDevice(0) != hipSuccess || hipMalloc(reinterpret_cast<void**>(&dev), most) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&pinned), most, hipHostMallocDefault) != hipSuccess) {
    fprintf(stderr, "no device / no memory\n");
    return 1;
  }
  char* plain = static_cast<char*>(aligned_alloc(4096, most));
  memset(plain, 3, most);
  memset(pinned, 5, most);
  const char* variant = getenv("FFX_DEBUG_VARIANT");
  printf("{\"variant\": \"%s\", \"rows\": [\n", variant ? variant : "");
  const size_t sizes[] = {size_t(6) << 20, size_t(12) << 20, size_t(25) << 20, size_t(50) << 20, size_t(75) << 20, size_t(256) << 20};
  bool first = true;
  for (size_t bytes : sizes) {
    double best[4] = {1e9, 1e9, 1e9, 1e9};  // staged in, staged out, page-locked in, page-locked out
    for (int rep = 0; rep < 6; ++rep) {
      double t0 = now();
      if (ffx_h2d_now(0, dev, plain, bytes) != 0) return 2;
      double t1 = now();
      if (ffx_d2h_now(0, plain, dev, bytes) != 0) return 2;
      double t2 = now();
      (void)hipMemcpy(dev, pinned, bytes, hipMemcpyHostToDevice);
      double t3 = now();
      (void)hipMemcpy(pinned, dev, bytes, hipMemcpyDeviceToHost);
      double t4 = now();
      if (rep == 0) continue;  // (blocks are made on the first transfer)
      best[0] = std::min(best[0], t1 - t0);
      best[1] = std::min(best[1], t2 - t1);
      best[2] = std::min(best[2], t3 - t2);
      best[3] = std::min(best[3], t4 - t3);
    }
    printf("%s  {\"MB\": %.0f, \"staged_in_GBps\": %.1f, \"staged_out_GBps\": %.1f, \"pinned_in_GBps\": %.1f, \"pinned_out_GBps\": %.1f, "
           "\"staged_in_ms\": %.3f, \"staged_out_ms\": %.3f, \"pinned_in_ms\": %.3f, \"pinned_out_ms\": %.3f}",
           first ? "" : ",\n", bytes / 1048576.0, bytes / best[0] / 1e9, bytes / best[1] / 1e9, bytes / best[2] / 1e9, bytes / best[3] / 1e9,
           best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3] * 1e3);
    first = false;
  }
  printf("\n],\n\"page_locked_in_pieces\": [\n");
  // the wire alone, cut the way the staging cuts it: 256 MB page-locked -> device as asynchronous copies of P MiB on one stream,
  // an event recorded behind each (what guards a block's reuse) or none
  hipStream_t st;
  (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipEvent_t evs[2];
  (void)hipEventCreateWithFlags(&evs[0], hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&evs[1], hipEventDisableTiming);
  first = true;
  for (int with_events = 0; with_events < 2; ++with_events)
    for (size_t piece_mb : {1, 2, 4, 8, 16, 32}) {
      const size_t piece = piece_mb << 20;
      double best_in = 1e9, best_out = 1e9;
      for (int rep = 0; rep < 4; ++rep) {
        for (int dir = 0; dir < 2; ++dir) {
          const double t0 = now();
          size_t k = 0;
          for (size_t at = 0; at < most; at += piece, ++k) {
            if (dir == 0) (void)hipMemcpyAsync(dev + at, pinned + at, piece, hipMemcpyHostToDevice, st);
            else (void)hipMemcpyAsync(pinned + at, dev + at, piece, hipMemcpyDeviceToHost, st);
            if (with_events) (void)hipEventRecord(evs[k & 1], st);
          }
          (void)hipStreamSynchronize(st);
          const double t = now() - t0;
          if (rep) (dir == 0 ? best_in : best_out) = std::min(dir == 0 ? best_in : best_out, t);
        }
      }
      printf("%s  {\"piece_MiB\": %zu, \"events\": %d, \"in_GBps\": %.1f, \"out_GBps\": %.1f}", first ? "" : ",\n", piece_mb, with_events,
             most / best_in / 1e9, most / best_out / 1e9);
      first = false;
    }
  printf("\n]}\n");
  return 0;
}
