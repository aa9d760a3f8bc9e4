// host_xfer.cpp -- see host_xfer.h: the caller's ordinary memory never reaches the runtime in pieces it would page-lock.
#include "host_xfer.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/ffx.h"
#include "../../include/ffx_private.h"
#include "ffx_internal.h"

int ffx_fail(int code, const char* fmt, ...);  // ffx.hip
extern thread_local std::string g_last_error;

namespace {

constexpr size_t kDirectBytes = size_t(1) << 20;  // at or below: the runtime copies through its own staging, it pins nothing
// (1 MiB pieces were tried -- less to fill before the wire is busy, less to drain at the end: the calls into the runtime, two per
// piece from four threads onto one stream, then cost more than the shorter fill saved: 1.9 instead of 1.4 ms for a 50 MB upload)
constexpr size_t kPiece = size_t(4) << 20;
constexpr int kLanes = 4;       // host threads a large transfer is spread over
constexpr int kLaneRoom = 8;    // (the experiments build can try more: xfer_lanes=N)
constexpr int kMaxDevices = 64;

#define X_HIP(expr)                                                                                              \
  do {                                                                                                           \
    hipError_t e_ = (expr);                                                                                      \
    if (e_ != hipSuccess)                                                                                        \
      return ffx_fail(e_ == hipErrorOutOfMemory ? FFX_OOM : FFX_HIP_ERROR, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                      __FILE__, __LINE__);                                                                       \
  } while (0)

struct Lane {
  char* block[2] = {nullptr, nullptr};
  hipEvent_t ev[2] = {nullptr, nullptr};  // block b's last copy (in: out of it, out: into it) is done
  bool busy[2] = {false, false};
  int make() {
    if (block[0]) return FFX_OK;
    for (int b = 0; b < 2; ++b) {
      X_HIP(hipEventCreateWithFlags(&ev[b], hipEventDisableTiming));
      X_HIP(hipHostMalloc(reinterpret_cast<void**>(&block[b]), kPiece, hipHostMallocDefault));
    }
    return FFX_OK;
  }
  void release() {
    for (int b = 1; b >= 0; --b) {
      if (ev[b] && busy[b]) (void)hipEventSynchronize(ev[b]);
      if (block[b]) (void)hipHostFree(block[b]);
      if (ev[b]) (void)hipEventDestroy(ev[b]);
      block[b] = nullptr;
      ev[b] = nullptr;
      busy[b] = false;
    }
  }
};

// one direction of one device: a transfer holds the mutex from its first piece to its last.  The pieces are queued on the
// CALLER'S stream by the lanes' threads (a stream takes calls from several threads); the lanes own no stream -- the device
// has four hardware queues for ordinary streams, and a copy stream that lands on the queue of the caller's kernel stream
// waits behind its kernels (measured: 3.2 ms instead of 0.1 ms for a 4 MB upload next to the scoring kernels).
struct Way {
  std::mutex mu;
  Lane lanes[kLaneRoom];
};

struct DeviceWays {
  Way in, out;
};

DeviceWays* g_ways[kMaxDevices] = {};
std::mutex g_ways_mu;

DeviceWays* ways_of(int device) {
  if (device < 0 || device >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lock(g_ways_mu);
  if (!g_ways[device]) g_ways[device] = new (std::nothrow) DeviceWays();
  return g_ways[device];
}

std::atomic<unsigned long long> g_staged_transfers{0}, g_staged_bytes{0};

// the piece size of a staged transfer.  Cutting small transfers finer (the first slices of an ffx_score call are 6 - 25 MB) so that
// the wire has something to do sooner was tried: 1 MiB pieces below 12 MB, 2 MiB below 32 MB gave 18.6 ms against 18.45 for the
// headline call with ordinary buffers -- the calls into the runtime cost what the earlier start saves (profiles/r05/xfer_piece_sweep.txt)
size_t piece_for(size_t) {
  if (long long v = 0; ffx_experiment("xfer_piece_kb", &v) && v >= 64 && (size_t)v * 1024 <= kPiece) return (size_t)v * 1024;
  return kPiece;
}

int lanes_for(size_t bytes, size_t piece) {
  const size_t pieces = (bytes + piece - 1) / piece;
  // two pieces per thread at least: starting one costs what copying ~1 MiB does
  size_t most = kLanes;
  if (long long v = 0; ffx_experiment("xfer_lanes", &v) && v >= 1 && v <= kLaneRoom) most = (size_t)v;
  return (int)std::max<size_t>(1, std::min<size_t>(most, pieces / 2));
}

template <class Body>
int on_threads(int T, Body&& body) {
  std::vector<int> rc((size_t)T, FFX_OK);
  std::vector<std::string> err((size_t)T);
  std::vector<std::thread> pool;
  auto run = [&](int t) {
    rc[(size_t)t] = body(t);
    if (rc[(size_t)t] != FFX_OK) err[(size_t)t] = g_last_error;  // thread-local: carried to the caller's thread
  };
  int started = 1;
  for (int t = 1; t < T; ++t) {
    try {
      pool.emplace_back(run, t);
      ++started;
    } catch (const std::system_error&) {
      break;  // no thread to be had: the calling thread takes the rest
    }
  }
  run(0);
  for (int t = started; t < T; ++t) run(t);
  for (auto& th : pool) th.join();
  for (int t = 0; t < T; ++t)
    if (rc[(size_t)t] != FFX_OK) {
      g_last_error = err[(size_t)t];
      return rc[(size_t)t];
    }
  return FFX_OK;
}

}  // namespace

bool ffx_runtime_owns(const void* p) {
  hipPointerAttribute_t attr;
  if (p && hipPointerGetAttributes(&attr, p) == hipSuccess)
    return attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
  (void)hipGetLastError();  // an ordinary pointer is "invalid value" to the runtime: not an error here
  return false;
}

int ffx_h2d_many(int device, int n_parts, void* const* dsts, const void* const* srcs, const size_t* sizes, hipStream_t stream) {
  // the parts that go to the runtime as they are (small, or page-locked by the caller), in order; the others are staged TOGETHER:
  // their pieces form one list that the lanes' threads share, so that several arrays of one batch (candidate ordinals and
  // first-stage scores of a slice) cost one round of threads and one piece's worth of filling before the wire is busy, not one each
  struct Piece {
    char* dst;
    const char* src;
    size_t len;
  };
  std::vector<Piece> pieces;
  std::vector<int> staged;  // the parts that travel through the blocks
  size_t staged_bytes = 0;
  for (int i = 0; i < n_parts; ++i) {
    const size_t bytes = sizes[i];
    if (bytes == 0) continue;
    if (!dsts[i] || !srcs[i]) return ffx_fail(FFX_BAD_ARG, "ffx_h2d: null pointer");
    if (bytes <= kDirectBytes || ffx_runtime_owns(srcs[i])) {
      X_HIP(hipMemcpyAsync(dsts[i], srcs[i], bytes, hipMemcpyHostToDevice, stream));
      continue;
    }
    staged.push_back(i);
    staged_bytes += bytes;
  }
  if (staged.empty()) return FFX_OK;
  const size_t piece = piece_for(staged_bytes);
  for (int i : staged)
    for (size_t at = 0; at < sizes[i]; at += piece)
      pieces.push_back({static_cast<char*>(dsts[i]) + at, static_cast<const char*>(srcs[i]) + at, std::min(piece, sizes[i] - at)});
  const int staged_parts = (int)staged.size();
  DeviceWays* dw = ways_of(device);
  if (!dw) return ffx_fail(FFX_BAD_ARG, "ffx_h2d: device %d", device);
  Way& w = dw->in;
  std::lock_guard<std::mutex> lock(w.mu);
  X_HIP(hipSetDevice(device));
  const int T = lanes_for(staged_bytes, piece);
  for (int t = 0; t < T; ++t) {
    const int rc = w.lanes[t].make();
    if (rc != FFX_OK) return rc;
  }
  const size_t n = pieces.size();
  const int rc = on_threads(T, [&](int t) -> int {
    Lane& ln = w.lanes[t];
    if (t > 0) X_HIP(hipSetDevice(device));
    int b = 0;
    for (size_t k = (size_t)t; k < n; k += (size_t)T, b ^= 1) {
      const Piece& pc = pieces[k];
      if (ln.busy[b]) X_HIP(hipEventSynchronize(ln.ev[b]));
      memcpy(ln.block[b], pc.src, pc.len);
      X_HIP(hipMemcpyAsync(pc.dst, ln.block[b], pc.len, hipMemcpyHostToDevice, stream));
      X_HIP(hipEventRecord(ln.ev[b], stream));
      ln.busy[b] = true;
    }
    return FFX_OK;
  });
  if (rc != FFX_OK) {
    // some pieces may be on their way out of the blocks with no event behind them (a failed record): wait them out here, so
    // that the blocks are free for the next transfer whatever the flags say
    (void)hipStreamSynchronize(stream);
    for (int t = 0; t < T; ++t) w.lanes[t].busy[0] = w.lanes[t].busy[1] = false;
    return rc;
  }
  g_staged_transfers.fetch_add((unsigned long long)staged_parts, std::memory_order_relaxed);
  g_staged_bytes.fetch_add(staged_bytes, std::memory_order_relaxed);
  return FFX_OK;
}

int ffx_h2d(int device, void* dst, const void* src, size_t bytes, hipStream_t stream) {
  return ffx_h2d_many(device, 1, &dst, &src, &bytes, stream);
}

int ffx_d2h(int device, void* dst, const void* src, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return FFX_OK;
  if (!dst || !src) return ffx_fail(FFX_BAD_ARG, "ffx_d2h: null pointer");
  if (bytes <= kDirectBytes || ffx_runtime_owns(dst)) {
    X_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
    return FFX_OK;
  }
  DeviceWays* dw = ways_of(device);
  if (!dw) return ffx_fail(FFX_BAD_ARG, "ffx_d2h: device %d", device);
  Way& w = dw->out;
  std::lock_guard<std::mutex> lock(w.mu);
  X_HIP(hipSetDevice(device));
  const int T = lanes_for(bytes, kPiece);
  for (int t = 0; t < T; ++t) {
    const int rc = w.lanes[t].make();
    if (rc != FFX_OK) return rc;
  }
  const size_t n = (bytes + kPiece - 1) / kPiece;
  const int rc = on_threads(T, [&](int t) -> int {
    Lane& ln = w.lanes[t];
    if (t > 0) X_HIP(hipSetDevice(device));
    auto issue = [&](size_t k, int b) -> int {
      const size_t at = k * kPiece, len = std::min(kPiece, bytes - at);
      X_HIP(hipMemcpyAsync(ln.block[b], static_cast<const char*>(src) + at, len, hipMemcpyDeviceToHost, stream));
      X_HIP(hipEventRecord(ln.ev[b], stream));
      ln.busy[b] = true;
      return FFX_OK;
    };
    // (the blocks of an out-lane are empty between transfers: every piece is copied out before the transfer returns)
    if ((size_t)t < n) {
      const int rc0 = issue((size_t)t, 0);
      if (rc0 != FFX_OK) return rc0;
    }
    int b = 0;
    for (size_t k = (size_t)t; k < n; k += (size_t)T, b ^= 1) {
      if (k + (size_t)T < n) {
        const int rc1 = issue(k + (size_t)T, b ^ 1);
        if (rc1 != FFX_OK) return rc1;
      }
      X_HIP(hipEventSynchronize(ln.ev[b]));
      ln.busy[b] = false;
      const size_t at = k * kPiece, len = std::min(kPiece, bytes - at);
      memcpy(static_cast<char*>(dst) + at, ln.block[b], len);
    }
    return FFX_OK;
  });
  if (rc != FFX_OK) {
    (void)hipStreamSynchronize(stream);  // nothing may still be landing in the blocks
    for (int t = 0; t < T; ++t) w.lanes[t].busy[0] = w.lanes[t].busy[1] = false;
    return rc;
  }
  g_staged_transfers.fetch_add(1, std::memory_order_relaxed);
  g_staged_bytes.fetch_add(bytes, std::memory_order_relaxed);
  return FFX_OK;
}

int ffx_h2d_now(int device, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return FFX_OK;
  X_HIP(hipSetDevice(device));
  const int rc = ffx_h2d(device, dst, src, bytes, nullptr);
  if (rc != FFX_OK) return rc;
  X_HIP(hipStreamSynchronize(nullptr));
  return FFX_OK;
}

int ffx_d2h_now(int device, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return FFX_OK;
  X_HIP(hipSetDevice(device));
  const int rc = ffx_d2h(device, dst, src, bytes, nullptr);
  if (rc != FFX_OK) return rc;
  X_HIP(hipStreamSynchronize(nullptr));
  return FFX_OK;
}

void ffx_xfer_release() {
  std::lock_guard<std::mutex> lock(g_ways_mu);
  for (int d = 0; d < kMaxDevices; ++d) {
    DeviceWays* dw = g_ways[d];
    if (!dw) continue;
    if (hipSetDevice(d) != hipSuccess) {
      (void)hipGetLastError();
      continue;
    }
    for (Way* w : {&dw->in, &dw->out}) {
      std::lock_guard<std::mutex> held(w->mu);
      for (auto& ln : w->lanes) ln.release();
    }
  }
}

void ffx_xfer_counters(unsigned long long* staged_transfers, unsigned long long* staged_bytes) {
  if (staged_transfers) *staged_transfers = g_staged_transfers.load();
  if (staged_bytes) *staged_bytes = g_staged_bytes.load();
}

extern "C" int ffx_host_xfer_stats(uint64_t* staged_transfers, uint64_t* staged_bytes) {
  unsigned long long t = 0, b = 0;
  ffx_xfer_counters(&t, &b);
  if (staged_transfers) *staged_transfers = t;
  if (staged_bytes) *staged_bytes = b;
  return FFX_OK;
}

extern "C" int ffx_host_xfer_release(void) {
  ffx_xfer_release();
  return FFX_OK;
}
