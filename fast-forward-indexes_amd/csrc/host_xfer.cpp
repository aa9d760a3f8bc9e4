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
// A copy command costs the wire ~15 us (+ ~5 us for the event behind it) whatever its size: page-locked memory cut into 4 MiB
// commands moves at 45.6 GB/s, 8 MiB 50.7, 16 MiB 53.4, uncut 57 (tools/probes/staged_xfer_probe.cpp) -- and a piece must be
// filled by a host thread before the wire can have it (4 MiB: 0.13 ms).  So pieces are GRADED: an upload begins with 1 MiB
// pieces (the wire is busy after 35 us) and grows to whole blocks; a download ends with small ones (the last piece's way out
// of its block is all that is not overlapped).  A fixed 4 MiB was the first form (45 GB/s at best; 1 MiB throughout: 30).
constexpr size_t kMiB = size_t(1) << 20;
constexpr size_t kBlock = 8 * kMiB;  // a lane's page-locked block = the largest piece
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
      X_HIP(hipHostMalloc(reinterpret_cast<void**>(&block[b]), kBlock, hipHostMallocDefault));
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

// length of piece k of a transfer counted from the graded end (uploads: its beginning, downloads: its end)
size_t graded_piece(size_t k) {
  if (long long v = 0; ffx_experiment("xfer_piece_kb", &v) && v >= 64 && (size_t)v * 1024 <= kBlock) return (size_t)v * 1024;  // a fixed size, for A/B
  return k < 2 ? kMiB : k < 4 ? 2 * kMiB : k < 6 ? 4 * kMiB : kBlock;
}

struct Piece {
  size_t part;  // which array of the call
  size_t at, len;
};

// the pieces of arrays of `sizes` bytes (those listed in `parts`), in the order they are to travel
std::vector<Piece> cut_graded(const std::vector<size_t>& parts, const size_t* sizes, bool small_first) {
  std::vector<Piece> pieces;
  size_t k = 0;
  if (small_first) {
    for (size_t i : parts)
      for (size_t at = 0; at < sizes[i]; ++k) {
        const size_t len = std::min(graded_piece(k), sizes[i] - at);
        pieces.push_back({i, at, len});
        at += len;
      }
  } else {  // the same cut made from the end, then turned round
    for (size_t r = parts.size(); r-- > 0;) {
      const size_t i = parts[r];
      for (size_t end = sizes[i]; end > 0; ++k) {
        const size_t len = std::min(graded_piece(k), end);
        pieces.push_back({i, end - len, len});
        end -= len;
      }
    }
    std::reverse(pieces.begin(), pieces.end());
  }
  return pieces;
}

int lanes_for(size_t n_pieces) {
  // two pieces per thread at least: starting one costs what copying ~1 MiB does
  size_t most = kLanes;
  if (long long v = 0; ffx_experiment("xfer_lanes", &v) && v >= 1 && v <= kLaneRoom) most = (size_t)v;
  return (int)std::max<size_t>(1, std::min<size_t>(most, n_pieces / 2));
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
  // first-stage scores of a slice) cost one round of threads and one small piece's worth of filling before the wire is busy
  std::vector<size_t> staged;  // the parts that travel through the blocks
  size_t staged_bytes = 0;
  for (int i = 0; i < n_parts; ++i) {
    const size_t bytes = sizes[i];
    if (bytes == 0) continue;
    if (!dsts[i] || !srcs[i]) return ffx_fail(FFX_BAD_ARG, "ffx_h2d: null pointer");
    if (bytes <= kDirectBytes || ffx_runtime_owns(srcs[i])) {
      X_HIP(hipMemcpyAsync(dsts[i], srcs[i], bytes, hipMemcpyHostToDevice, stream));
      continue;
    }
    staged.push_back((size_t)i);
    staged_bytes += bytes;
  }
  if (staged.empty()) return FFX_OK;
  const std::vector<Piece> pieces = cut_graded(staged, sizes, /*small_first=*/true);
  DeviceWays* dw = ways_of(device);
  if (!dw) return ffx_fail(FFX_BAD_ARG, "ffx_h2d: device %d", device);
  Way& w = dw->in;
  std::lock_guard<std::mutex> lock(w.mu);
  X_HIP(hipSetDevice(device));
  const int T = lanes_for(pieces.size());
  for (int t = 0; t < T; ++t) {
    const int rc = w.lanes[t].make();
    if (rc != FFX_OK) return rc;
  }
  std::atomic<size_t> next{0};  // the threads take the pieces as they come free: the first ones go to whoever is up first
  const int rc = on_threads(T, [&](int t) -> int {
    Lane& ln = w.lanes[t];
    if (t > 0) X_HIP(hipSetDevice(device));
    int b = 0;
    for (size_t k; (k = next.fetch_add(1, std::memory_order_relaxed)) < pieces.size(); b ^= 1) {
      const Piece& pc = pieces[k];
      if (ln.busy[b]) X_HIP(hipEventSynchronize(ln.ev[b]));
      memcpy(ln.block[b], static_cast<const char*>(srcs[pc.part]) + pc.at, pc.len);
      X_HIP(hipMemcpyAsync(static_cast<char*>(dsts[pc.part]) + pc.at, ln.block[b], pc.len, hipMemcpyHostToDevice, stream));
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
  g_staged_transfers.fetch_add((unsigned long long)staged.size(), std::memory_order_relaxed);
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
  const std::vector<Piece> pieces = cut_graded({0}, &bytes, /*small_first=*/false);
  const size_t n = pieces.size();
  const int T = lanes_for(n);
  for (int t = 0; t < T; ++t) {
    const int rc = w.lanes[t].make();
    if (rc != FFX_OK) return rc;
  }
  std::atomic<size_t> next{0};
  const int rc = on_threads(T, [&](int t) -> int {
    Lane& ln = w.lanes[t];
    if (t > 0) X_HIP(hipSetDevice(device));
    auto issue = [&](size_t k, int b) -> int {
      X_HIP(hipMemcpyAsync(ln.block[b], static_cast<const char*>(src) + pieces[k].at, pieces[k].len, hipMemcpyDeviceToHost, stream));
      X_HIP(hipEventRecord(ln.ev[b], stream));
      ln.busy[b] = true;
      return FFX_OK;
    };
    // (the blocks of an out-lane are empty between transfers: every piece is copied out before the transfer returns)
    // a thread has one piece landing in one block while it empties the other
    size_t cur = next.fetch_add(1, std::memory_order_relaxed);
    int b = 0;
    if (cur < n) {
      const int rc0 = issue(cur, b);
      if (rc0 != FFX_OK) return rc0;
    }
    while (cur < n) {
      const size_t nxt = next.fetch_add(1, std::memory_order_relaxed);
      if (nxt < n) {
        const int rc1 = issue(nxt, b ^ 1);
        if (rc1 != FFX_OK) return rc1;
      }
      X_HIP(hipEventSynchronize(ln.ev[b]));
      ln.busy[b] = false;
      memcpy(static_cast<char*>(dst) + pieces[cur].at, ln.block[b], pieces[cur].len);
      cur = nxt;
      b ^= 1;
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
