// comm.hip -- multi-GPU half of the C ABI (include/ffx.h: ffx_comm_*, ffx_route, ffx_sharded_*).
//
// One process per GPU.  The index is row-sharded by document; every rank scores the candidates
// whose documents it holds and the dense scores of a query travel to the rank that owns the
// query: ONE all-to-all-v of fp32 scores per batch.  Two transports sit behind the
// same entry points:
//   FFX_COMM_RCCL  ncclSend/ncclRecv full mesh inside one ncclGroup on the communicator's HIP
//                  stream -- every peer slice rides its own xGMI link.  librccl is dlopen'ed on
//                  first use, so a single-GPU process never loads it.
//   FFX_COMM_HOST  the same exchange staged through POSIX shared memory: ranks may share a GPU
//                  (RCCL refuses that) or have none at all (host buffers; the CPU test-suite).
// Nothing here depends on PyTorch; this file uses the public ABI of ffx.hip only.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is resolved with dlopen/dlsym

#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "../../include/ffx.h"
#include "../../include/ffx_private.h"
#include "host_xfer.h"
#include "ffx_internal.h"  // ffx_fail, ffx_debug_variant, in-place hooks of plans and merges (ffx.hip)

namespace {

#define C_HIP(expr)                                                                                \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return ffx_fail(e_ == hipErrorOutOfMemory ? FFX_OOM : FFX_HIP_ERROR, "%s: %s (%s:%d)", #expr, \
                      hipGetErrorString(e_), __FILE__, __LINE__);                                  \
  } while (0)

#define C_TRY(expr)                \
  do {                             \
    int rc_ = (expr);              \
    if (rc_ != FFX_OK) return rc_; \
  } while (0)

// ---- librccl, resolved lazily ---------------------------------------------------------------------
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  std::string error;
};

Rccl* rccl() {
  static Rccl r;
  static bool tried = false;
  if (tried) return &r;
  tried = true;
  // the soname first: if another library of this process already brought an RCCL (and with it a
  // HIP runtime), that one must be used
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (r.handle) break;
  }
  if (!r.handle) {
    r.error = std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "?");
    return &r;
  }
  bool ok = true;
#define SYM(field, name)                                                \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.handle, name)); \
  if (!r.field) {                                                       \
    ok = false;                                                         \
    r.error = std::string("librccl lacks ") + name;                     \
  }
  SYM(GetUniqueId, "ncclGetUniqueId")
  SYM(CommInitRank, "ncclCommInitRank")
  SYM(CommDestroy, "ncclCommDestroy")
  SYM(GetErrorString, "ncclGetErrorString")
  SYM(GroupStart, "ncclGroupStart")
  SYM(GroupEnd, "ncclGroupEnd")
  SYM(Send, "ncclSend")
  SYM(Recv, "ncclRecv")
  SYM(AllReduce, "ncclAllReduce")
  SYM(AllGather, "ncclAllGather")
#undef SYM
  if (!ok) {
    dlclose(r.handle);
    r.handle = nullptr;
  }
  return &r;
}

#define C_NCCL(expr)                                                                                        \
  do {                                                                                                      \
    ncclResult_t r_ = (expr);                                                                               \
    if (r_ != ncclSuccess)                                                                                  \
      return ffx_fail(FFX_RCCL_ERROR, "%s: %s (%s:%d)", #expr, rccl()->GetErrorString(r_), __FILE__, __LINE__); \
  } while (0)

// ---- shared-memory transport ----------------------------------------------------------------------
constexpr int kMaxWorld = 64;
constexpr size_t kSmall = 65536;      // bytes of control payload per rank and round
constexpr uint32_t kMagic = 0x46465831;  // "FFX1"

struct ShmSlot {
  std::atomic<uint32_t> data_gen;  // generation of this rank's data segment (0: none yet)
  uint32_t pad;
  uint64_t data_cap;               // its size in bytes
  int64_t counts[kMaxWorld];       // elements this rank sends to each peer in the current exchange
  char small[kSmall];
};

struct ShmHeader {
  std::atomic<uint32_t> magic;
  uint32_t world;
  std::atomic<uint32_t> arrived, generation;  // sense-counting barrier
  std::atomic<uint32_t> failed;               // a rank gave up: everybody waiting fails too
  uint32_t pad;
  ShmSlot slot[1];  // [world]
};

double barrier_timeout_s() {
  const char* e = getenv("FFX_COMM_TIMEOUT_S");
  const double v = e ? atof(e) : 0.0;
  return v > 0 ? v : 120.0;
}

struct Mapping {
  void* p = nullptr;
  size_t bytes = 0;
  uint32_t gen = 0;
  void unmap() {
    if (p) munmap(p, bytes);
    p = nullptr;
    bytes = 0;
  }
};

}  // namespace

struct ffx_comm {
  int transport = FFX_COMM_RCCL, device = -1, rank = 0, world = 1;
  hipStream_t stream = nullptr;
  // RCCL
  ncclComm_t nccl = nullptr;
  void* ctl_dev = nullptr;   // kSmall * (world + 1) bytes of device scratch for the control collectives
  void* ctl_host = nullptr;  // page-locked mirror
  float* stage_dev = nullptr;  // ffx_comm_exchange_host over RCCL: device staging, grown as needed and kept
  size_t stage_cap = 0;        // (floats)
  // The ROUTING lane (ffx_sharded_stage_owned): candidate ids of batch i+1 travel from the ranks that own its queries to
  // the ranks that hold the rows while batch i is being scored.  The score exchange of batch i is queued on `stream`
  // BEHIND batch i's gather_dot (an event wait), so anything queued on that stream afterwards would wait too; the
  // routing traffic therefore has a stream of its own and, over RCCL, a communicator of its own (operations of one
  // communicator execute in issue order).  Created on first use (collective).
  ncclComm_t nccl_route = nullptr;
  hipStream_t route_stream = nullptr;
  // rehearsal: what stands in for the peers' memory -- a zeroed source the "received" pieces come out of (grown as needed);
  // the "sent" pieces are read and stored nowhere (the peer's memory is not on this GPU)
  float* reh_src = nullptr;
  size_t reh_cap = 0;
  // rehearsal of the routing lane: what the peers WOULD have sent (made by the caller, who knows the whole batch),
  // consumed piece by piece by the exchanges of that lane; zeros once it is used up (ffx_comm_rehearsal_inbox)
  const uint32_t* inbox = nullptr;
  size_t inbox_words = 0, inbox_at = 0;
  // shared memory
  std::string name;
  ShmHeader* hdr = nullptr;
  size_t hdr_bytes = 0;
  Mapping seg[kMaxWorld];  // seg[rank] is this rank's own send segment
};

namespace {

size_t header_bytes(int world) { return sizeof(ShmHeader) + sizeof(ShmSlot) * (size_t)(world - 1); }

std::string seg_name(const ffx_comm* c, int rank, uint32_t gen) {
  char buf[160];
  snprintf(buf, sizeof buf, "%s.r%d.g%u", c->name.c_str(), rank, gen);
  return buf;
}

int shm_barrier(ffx_comm* c) {
  ShmHeader* h = c->hdr;
  if (h->failed.load()) return ffx_fail(FFX_RCCL_ERROR, "host transport: a peer rank failed");
  const uint32_t gen = h->generation.load();
  if (h->arrived.fetch_add(1) + 1 == (uint32_t)c->world) {
    h->arrived.store(0);
    h->generation.fetch_add(1);
    return FFX_OK;
  }
  const auto t0 = std::chrono::steady_clock::now();
  const double limit = barrier_timeout_s();
  for (unsigned spin = 0; h->generation.load() == gen; ++spin) {
    if (h->failed.load()) return ffx_fail(FFX_RCCL_ERROR, "host transport: a peer rank failed");
    if (spin < 2000) {
      sched_yield();
    } else {
      usleep(50);
      if ((spin & 1023) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
        h->failed.store(1);
        return ffx_fail(FFX_RCCL_ERROR, "host transport: rank %d waited %.0f s at a barrier (peer gone?)", c->rank, limit);
      }
    }
  }
  return FFX_OK;
}

// this rank's send segment must hold `bytes`
int shm_own_segment(ffx_comm* c, size_t bytes) {
  Mapping& m = c->seg[c->rank];
  if (m.p && m.bytes >= bytes) return FFX_OK;
  const uint32_t gen = m.gen + 1;
  const size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 16);
  const std::string nm = seg_name(c, c->rank, gen);
  int fd = shm_open(nm.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return ffx_fail(FFX_OOM, "shm_open(%s) failed: %s", nm.c_str(), strerror(errno));
  if (ftruncate(fd, (off_t)want) != 0) {
    close(fd);
    shm_unlink(nm.c_str());
    return ffx_fail(FFX_OOM, "ftruncate(%s, %zu) failed: %s", nm.c_str(), want, strerror(errno));
  }
  void* p = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    shm_unlink(nm.c_str());
    return ffx_fail(FFX_OOM, "mmap(%s) failed: %s", nm.c_str(), strerror(errno));
  }
  if (m.p) {  // peers re-map when they see the new generation; the old pages go with the last unmap
    shm_unlink(seg_name(c, c->rank, m.gen).c_str());
    m.unmap();
  }
  m.p = p;
  m.bytes = want;
  m.gen = gen;
  c->hdr->slot[c->rank].data_cap = want;
  c->hdr->slot[c->rank].data_gen.store(gen);
  return FFX_OK;
}

int shm_peer_segment(ffx_comm* c, int peer) {
  if (peer == c->rank) return FFX_OK;
  const uint32_t gen = c->hdr->slot[peer].data_gen.load();
  Mapping& m = c->seg[peer];
  if (m.p && m.gen == gen) return FFX_OK;
  m.unmap();
  if (gen == 0) return FFX_OK;
  const std::string nm = seg_name(c, peer, gen);
  int fd = shm_open(nm.c_str(), O_RDONLY, 0);
  if (fd < 0) return ffx_fail(FFX_RCCL_ERROR, "shm_open(%s) failed: %s", nm.c_str(), strerror(errno));
  const size_t bytes = c->hdr->slot[peer].data_cap;
  void* p = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ffx_fail(FFX_RCCL_ERROR, "mmap(%s) failed: %s", nm.c_str(), strerror(errno));
  m.p = p;
  m.bytes = bytes;
  m.gen = gen;
  return FFX_OK;
}

// all-gather of <= kSmall bytes per rank through the control slots
int shm_allgather_small(ffx_comm* c, const void* mine, size_t bytes, void* all) {
  memcpy(c->hdr->slot[c->rank].small, mine, bytes);
  C_TRY(shm_barrier(c));
  for (int r = 0; r < c->world; ++r) memcpy(static_cast<char*>(all) + (size_t)r * bytes, c->hdr->slot[r].small, bytes);
  return shm_barrier(c);
}

int rccl_allgather_small(ffx_comm* c, const void* mine, size_t bytes, void* all) {
  char* host = static_cast<char*>(c->ctl_host);
  char* dev = static_cast<char*>(c->ctl_dev);
  memcpy(host, mine, bytes);
  C_HIP(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
  C_NCCL(rccl()->AllGather(dev, dev + kSmall, bytes, ncclChar, c->nccl, c->stream));
  C_HIP(hipMemcpyAsync(host + kSmall, dev + kSmall, bytes * c->world, hipMemcpyDeviceToHost, c->stream));
  C_HIP(hipStreamSynchronize(c->stream));
  memcpy(all, host + kSmall, bytes * c->world);
  return FFX_OK;
}

int allgather_small(ffx_comm* c, const void* mine, size_t bytes, void* all) {
  if (c->world == 1) {
    memcpy(all, mine, bytes);
    return FFX_OK;
  }
  if (c->transport == FFX_COMM_REHEARSAL) {  // the peers do not exist: every slot holds this rank's contribution
    for (int r = 0; r < c->world; ++r) memcpy(static_cast<char*>(all) + (size_t)r * bytes, mine, bytes);
    return FFX_OK;
  }
  return c->transport == FFX_COMM_RCCL ? rccl_allgather_small(c, mine, bytes, all) : shm_allgather_small(c, mine, bytes, all);
}

void destroy_comm(ffx_comm* c) {
  if (!c) return;
  if (c->device >= 0) (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->route_stream) (void)hipStreamSynchronize(c->route_stream);
  if (c->nccl_route) (void)rccl()->CommDestroy(c->nccl_route);
  if (c->nccl) (void)rccl()->CommDestroy(c->nccl);
  if (c->ctl_dev) (void)hipFree(c->ctl_dev);
  if (c->ctl_host) (void)hipHostFree(c->ctl_host);
  if (c->stage_dev) (void)hipFree(c->stage_dev);
  if (c->reh_src) (void)hipFree(c->reh_src);
  if (c->hdr) {
    if (c->seg[c->rank].p) shm_unlink(seg_name(c, c->rank, c->seg[c->rank].gen).c_str());
    for (int r = 0; r < c->world; ++r) c->seg[r].unmap();
    // (rank 0 unlinked the header's name right after start-up, when everybody had attached)
    munmap(c->hdr, c->hdr_bytes);
  }
  if (c->route_stream) (void)hipStreamDestroy(c->route_stream);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

}  // namespace

extern "C" {

int ffx_comm_unique_id(int transport, void* id) {
  if (!id) return ffx_fail(FFX_BAD_ARG, "id is NULL");
  memset(id, 0, FFX_COMM_ID_BYTES);
  if (transport == FFX_COMM_RCCL) {
    static_assert(sizeof(ncclUniqueId) == FFX_COMM_ID_BYTES, "FFX_COMM_ID_BYTES must hold an ncclUniqueId");
    Rccl* r = rccl();
    if (!r->handle) return ffx_fail(FFX_RCCL_ERROR, "%s", r->error.c_str());
    ncclUniqueId uid;
    C_NCCL(r->GetUniqueId(&uid));
    memcpy(id, &uid, sizeof uid);
    return FFX_OK;
  }
  if (transport == FFX_COMM_HOST) {  // a shared-memory name nobody else will pick
    unsigned char rnd[16];
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f || fread(rnd, 1, sizeof rnd, f) != sizeof rnd) {
      if (f) fclose(f);
      return ffx_fail(FFX_RCCL_ERROR, "cannot read /dev/urandom");
    }
    fclose(f);
    char* out = static_cast<char*>(id);
    int n = snprintf(out, FFX_COMM_ID_BYTES, "/ffx.%d.", (int)getpid());
    for (unsigned char b : rnd) n += snprintf(out + n, FFX_COMM_ID_BYTES - n, "%02x", b);
    return FFX_OK;
  }
  if (transport == FFX_COMM_REHEARSAL) return FFX_OK;  // nobody to hand it to
  return ffx_fail(FFX_BAD_ARG, "unknown transport %d", transport);
}

int ffx_comm_init(int transport, int device, int rank, int world, const void* id, ffx_comm** out) {
  if (!out || !id) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  *out = nullptr;
  if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world) return ffx_fail(FFX_BAD_ARG, "bad rank %d / world %d (at most %d ranks)", rank, world, kMaxWorld);
  if (transport != FFX_COMM_RCCL && transport != FFX_COMM_HOST && transport != FFX_COMM_REHEARSAL) return ffx_fail(FFX_BAD_ARG, "unknown transport %d", transport);
  if (transport != FFX_COMM_HOST && device < 0) return ffx_fail(FFX_BAD_ARG, "the %s transport needs a device", transport == FFX_COMM_RCCL ? "RCCL" : "rehearsal");
  ffx_comm* c = new (std::nothrow) ffx_comm();
  if (!c) return ffx_fail(FFX_OOM, "host allocation failed");
  c->transport = transport;
  c->device = device;
  c->rank = rank;
  c->world = world;
  int rc = FFX_OK;
  auto bail = [&](int code) {
    destroy_comm(c);
    return code;
  };
  if (device >= 0) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
      (void)hipGetLastError();
      return bail(ffx_fail(FFX_NO_DEVICE, "no HIP device: libffx has no CPU fallback"));
    }
    if (device >= n) return bail(ffx_fail(FFX_BAD_ARG, "device %d out of range (%d devices)", device, n));
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess)
      return bail(ffx_fail(FFX_HIP_ERROR, "stream creation failed on device %d", device));
  }
  if (transport == FFX_COMM_RCCL) {
    Rccl* r = rccl();
    if (!r->handle) return bail(ffx_fail(FFX_RCCL_ERROR, "%s", r->error.c_str()));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclResult_t nr = r->CommInitRank(&c->nccl, world, uid, rank);
    if (nr != ncclSuccess) {
      c->nccl = nullptr;
      return bail(ffx_fail(FFX_RCCL_ERROR, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, device, r->GetErrorString(nr)));
    }
    const size_t ctl = kSmall * (size_t)(world + 1);
    if (hipMalloc(&c->ctl_dev, ctl) != hipSuccess || hipHostMalloc(&c->ctl_host, ctl, hipHostMallocDefault) != hipSuccess)
      return bail(ffx_fail(FFX_OOM, "control buffers"));
  } else if (transport == FFX_COMM_HOST) {
    const char* nm = static_cast<const char*>(id);
    if (nm[0] != '/' || memchr(nm, 0, FFX_COMM_ID_BYTES) == nullptr) return bail(ffx_fail(FFX_BAD_ARG, "not an id of ffx_comm_unique_id(FFX_COMM_HOST)"));
    c->name = nm;
    c->hdr_bytes = header_bytes(world);
    int fd = -1;
    if (rank == 0) {
      fd = shm_open(nm, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0) return bail(ffx_fail(FFX_RCCL_ERROR, "shm_open(%s) failed: %s", nm, strerror(errno)));
      if (ftruncate(fd, (off_t)c->hdr_bytes) != 0) {
        close(fd);
        shm_unlink(nm);
        return bail(ffx_fail(FFX_OOM, "ftruncate(%s) failed: %s", nm, strerror(errno)));
      }
    } else {  // wait for rank 0 to create and size the segment
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        fd = shm_open(nm, O_RDWR, 0);
        struct stat sb;
        if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= c->hdr_bytes) break;
        if (fd >= 0) close(fd);
        fd = -1;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > barrier_timeout_s())
          return bail(ffx_fail(FFX_RCCL_ERROR, "rank %d: rank 0 never created %s", rank, nm));
        usleep(1000);
      }
    }
    void* p = mmap(nullptr, c->hdr_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return bail(ffx_fail(FFX_OOM, "mmap(%s) failed: %s", nm, strerror(errno)));
    c->hdr = static_cast<ShmHeader*>(p);
    if (rank == 0) {  // fresh pages are zero: counters start at 0
      c->hdr->world = (uint32_t)world;
      c->hdr->magic.store(kMagic);
    } else {
      const auto t0 = std::chrono::steady_clock::now();
      while (c->hdr->magic.load() != kMagic) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > barrier_timeout_s())
          return bail(ffx_fail(FFX_RCCL_ERROR, "rank %d: %s was never initialised", rank, nm));
        usleep(200);
      }
      if (c->hdr->world != (uint32_t)world) return bail(ffx_fail(FFX_BAD_ARG, "rank %d expects %d ranks, rank 0 set up %u", rank, world, c->hdr->world));
    }
    rc = shm_barrier(c);  // everybody is attached: the name is no longer needed
    if (rank == 0) shm_unlink(nm);
    if (rc != FFX_OK) return bail(rc);
  }
  *out = c;
  return FFX_OK;
}

int ffx_comm_destroy(ffx_comm* c) {
  destroy_comm(c);
  return FFX_OK;
}

int ffx_comm_rank(const ffx_comm* c, int* rank, int* world) {
  if (!c) return ffx_fail(FFX_BAD_ARG, "NULL comm");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return FFX_OK;
}

int ffx_comm_stream(ffx_comm* c, void** stream) {
  if (!c || !stream) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  *stream = (void*)c->stream;
  return FFX_OK;
}

int ffx_comm_allgather(ffx_comm* c, const void* mine, int64_t bytes, void* all) {
  if (!c || bytes < 0 || (bytes > 0 && (!mine || !all))) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (c->device >= 0) C_HIP(hipSetDevice(c->device));
  for (int64_t at = 0; at < bytes; at += (int64_t)kSmall) {  // control-sized messages: a few rounds at most
    const size_t n = (size_t)std::min<int64_t>((int64_t)kSmall, bytes - at);
    std::vector<char> round(n * (size_t)c->world);
    C_TRY(allgather_small(c, static_cast<const char*>(mine) + at, n, round.data()));
    for (int r = 0; r < c->world; ++r) memcpy(static_cast<char*>(all) + (size_t)r * (size_t)bytes + (size_t)at, round.data() + (size_t)r * n, n);
  }
  return FFX_OK;
}

int ffx_comm_barrier(ffx_comm* c) {
  if (!c) return ffx_fail(FFX_BAD_ARG, "NULL comm");
  if (c->world == 1 || c->transport == FFX_COMM_REHEARSAL) return FFX_OK;
  if (c->transport == FFX_COMM_HOST) return shm_barrier(c);
  char one = 1;
  std::vector<char> all((size_t)c->world);
  C_HIP(hipSetDevice(c->device));
  return rccl_allgather_small(c, &one, 1, all.data());
}

int ffx_comm_allreduce_f64(ffx_comm* c, double* inout, int32_t n, int op) {
  if (!c || n < 0 || (n > 0 && !inout)) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (op < FFX_OP_SUM || op > FFX_OP_MIN) return ffx_fail(FFX_BAD_ARG, "unknown reduction %d", op);
  if (n == 0 || c->world == 1) return FFX_OK;
  std::vector<double> all((size_t)n * (size_t)c->world);
  C_TRY(ffx_comm_allgather(c, inout, (int64_t)n * 8, all.data()));
  for (int i = 0; i < n; ++i) {
    double v = all[i];
    for (int r = 1; r < c->world; ++r) {
      const double x = all[(size_t)r * n + i];
      v = op == FFX_OP_SUM ? v + x : op == FFX_OP_MAX ? std::max(v, x) : std::min(v, x);
    }
    inout[i] = v;
  }
  return FFX_OK;
}

// ---- the all-to-all-v underneath every exchange ----------------------------------------------------------------------
// One piece per peer, in 4-byte words (fp32 scores, int32 candidate ordinals, per-query counts, query vectors: nothing is
// ever reduced on the way, so the word is the unit): what this rank sends to the peer and what it receives from it.
// Two lanes: 0 = the score exchange (communicator's stream), 1 = the routing lane (its own stream and, over RCCL, its
// own communicator: see ffx_comm).
struct Piece {
  const void* send = nullptr;
  int64_t n_send = 0;
  void* recv = nullptr;
  int64_t n_recv = 0;
};

static int check_pieces(const ffx_comm* c, const Piece* pc, int64_t* n_send, int64_t* n_recv) {
  *n_send = *n_recv = 0;
  for (int r = 0; r < c->world; ++r) {
    if (pc[r].n_send < 0 || pc[r].n_recv < 0) return ffx_fail(FFX_BAD_ARG, "negative count for peer %d", r);
    if ((pc[r].n_send > 0 && !pc[r].send) || (pc[r].n_recv > 0 && !pc[r].recv)) return ffx_fail(FFX_BAD_ARG, "NULL buffer for peer %d", r);
    *n_send += pc[r].n_send;
    *n_recv += pc[r].n_recv;
  }
  if (pc[c->rank].n_send != pc[c->rank].n_recv)
    return ffx_fail(FFX_BAD_ARG, "a rank sends itself %lld and expects %lld", (long long)pc[c->rank].n_send, (long long)pc[c->rank].n_recv);
  return FFX_OK;
}

static int check_counts(const ffx_comm* c, const int64_t* send_counts, const int64_t* recv_counts) {
  if (!send_counts || !recv_counts) return ffx_fail(FFX_BAD_ARG, "counts are NULL");
  return FFX_OK;
}

// contiguous buffers with the slices in rank order (the public form) -> pieces
static void pieces_of(const ffx_comm* c, const float* send, const int64_t* send_counts, float* recv, const int64_t* recv_counts, Piece* pc) {
  int64_t s_at = 0, r_at = 0;
  for (int r = 0; r < c->world; ++r) {
    pc[r].send = send ? send + s_at : nullptr;
    pc[r].n_send = send_counts[r];
    pc[r].recv = recv ? recv + r_at : nullptr;
    pc[r].n_recv = recv_counts[r];
    s_at += send_counts[r];
    r_at += recv_counts[r];
  }
}

// host transport: publish what this rank sends (packed in rank order into its segment), then pull this rank's piece out
// of every peer's segment.  `device`: the pieces are device pointers (copies ordered on `st`); else plain host memory.
// Once this rank's counts are published, EVERY failure marks the communicator failed before returning, so that
// the peers waiting at a barrier fail promptly instead of spinning out their timeout.
static int shm_exchange_body(ffx_comm* c, const Piece* pc, bool device, hipStream_t st) {
  int64_t n_send = 0, n_recv = 0;
  C_TRY(check_pieces(c, pc, &n_send, &n_recv));
  C_TRY(shm_own_segment(c, (size_t)n_send * 4));
  ShmSlot& mine = c->hdr->slot[c->rank];
  for (int r = 0; r < c->world; ++r) mine.counts[r] = pc[r].n_send;
  int64_t at = 0;
  for (int r = 0; r < c->world; ++r) {
    if (pc[r].n_send > 0) {
      char* dst = static_cast<char*>(c->seg[c->rank].p) + (size_t)at * 4;
      if (device)
        C_TRY(ffx_d2h(c->device, dst, pc[r].send, (size_t)pc[r].n_send * 4, st));
      else
        memcpy(dst, pc[r].send, (size_t)pc[r].n_send * 4);
    }
    at += pc[r].n_send;
  }
  if (device && n_send > 0) C_HIP(hipStreamSynchronize(st));
  C_TRY(shm_barrier(c));  // every send segment is complete
  for (int r = 0; r < c->world; ++r) {
    const ShmSlot& peer = c->hdr->slot[r];
    if (peer.counts[c->rank] != pc[r].n_recv)
      return ffx_fail(FFX_BAD_ARG, "rank %d sends %lld words to rank %d, which expects %lld", r, (long long)peer.counts[c->rank], c->rank, (long long)pc[r].n_recv);
    if (pc[r].n_recv > 0) {
      C_TRY(shm_peer_segment(c, r));
      int64_t off = 0;
      for (int i = 0; i < c->rank; ++i) {
        if (peer.counts[i] < 0) return ffx_fail(FFX_RCCL_ERROR, "host transport: rank %d published a negative count", r);
        off += peer.counts[i];
      }
      // what the peer claims to send must lie inside the segment this rank mapped (counts come from shared memory)
      if (r != c->rank && ((uint64_t)off + (uint64_t)pc[r].n_recv) * 4 > c->seg[r].bytes)
        return ffx_fail(FFX_RCCL_ERROR, "host transport: rank %d's piece for rank %d lies outside its %zu-byte segment", r, c->rank, c->seg[r].bytes);
      const char* src = static_cast<const char*>(c->seg[r].p) + (size_t)off * 4;
      const size_t bytes = (size_t)pc[r].n_recv * 4;
      if (device)
        C_TRY(ffx_h2d(c->device, pc[r].recv, src, bytes, st));
      else
        memcpy(pc[r].recv, src, bytes);
    }
  }
  if (device) C_HIP(hipStreamSynchronize(st));
  return shm_barrier(c);  // every rank has read: the segments may be overwritten
}

static int shm_exchange(ffx_comm* c, const Piece* pc, bool device, hipStream_t st) {
  const int rc = shm_exchange_body(c, pc, device, st);
  if (rc != FFX_OK) c->hdr->failed.store(1);
  return rc;
}

// rehearsal: the bytes of one exchange through this GPU's memory: piece i copies n[i] words from src[i] to dst[i]
struct RehearsalPieces {
  const uint32_t* src[2 * kMaxWorld];
  uint32_t* dst[2 * kMaxWorld];
  long n[2 * kMaxWorld];
  int count;
  // paced form: ns per KiB of a piece (what a link would need for it), and a flag word to poll while waiting
  float ns_per_kib;
  const uint32_t* flag;
};

// n words from s to d by `nthreads` threads: 16-byte stores, 16-byte loads where the source allows, four of them in flight
// per thread (a copy loop that keeps about as many bytes in flight per workgroup as a collective library's does; with
// 4-byte accesses the stand-in is latency-bound beside an HBM-saturating kernel and says more about itself than about
// the step it rehearses)
__device__ inline void copy_words(const uint32_t* __restrict__ s, uint32_t* __restrict__ d, long n, long tid, long nthreads) {
  if (!d) {  // a piece that leaves this GPU: it is read here and written elsewhere
    uint32_t acc = 0;
    long head = (long)(((16 - ((uintptr_t)s & 15)) & 15) >> 2);
    if (head > n) head = n;
    for (long e = tid; e < head; e += nthreads) acc ^= s[e];
    const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(s + head);
    const long n4 = (n - head) >> 2;
    long e = tid;
    for (; e + 3 * nthreads < n4; e += 4 * nthreads) {
      const uint4 a = s4[e], b = s4[e + nthreads], c = s4[e + 2 * nthreads], f = s4[e + 3 * nthreads];
      acc ^= a.x ^ a.w ^ b.x ^ b.w ^ c.x ^ c.w ^ f.x ^ f.w;
    }
    for (; e < n4; e += nthreads) acc ^= s4[e].x ^ s4[e].w;
    for (long e2 = head + (n4 << 2) + tid; e2 < n; e2 += nthreads) acc ^= s[e2];
    if (acc == 0x9E3779B9u && n < 0) const_cast<uint32_t*>(s)[0] = acc;  // (never: keeps the loads)
    return;
  }
  long head = (long)(((16 - ((uintptr_t)d & 15)) & 15) >> 2);
  if (head > n) head = n;
  for (long e = tid; e < head; e += nthreads) d[e] = s[e];
  s += head;
  d += head;
  n -= head;
  const long n4 = n >> 2;
  uint4* __restrict__ d4 = reinterpret_cast<uint4*>(d);
  if (((uintptr_t)s & 15) == 0) {
    const uint4* __restrict__ s4 = reinterpret_cast<const uint4*>(s);
    long e = tid;
    for (; e + 3 * nthreads < n4; e += 4 * nthreads) {
      const uint4 a = s4[e], b = s4[e + nthreads], c = s4[e + 2 * nthreads], f = s4[e + 3 * nthreads];
      d4[e] = a;
      d4[e + nthreads] = b;
      d4[e + 2 * nthreads] = c;
      d4[e + 3 * nthreads] = f;
    }
    for (; e < n4; e += nthreads) d4[e] = s4[e];
  } else {
    for (long e = tid; e < n4; e += nthreads) {
      uint4 v;
      v.x = s[4 * e];
      v.y = s[4 * e + 1];
      v.z = s[4 * e + 2];
      v.w = s[4 * e + 3];
      d4[e] = v;
    }
  }
  for (long e = (n4 << 2) + tid; e < n; e += nthreads) d[e] = s[e];
}

// ONE kernel on a handful of workgroups (RCCL groups its sends and receives into one kernel on a handful of workgroups too)
// (32 registers per lane: next to gather_dot's six waves per SIMD that is what is left -- a kernel that needs more is not
// scheduled until gather_dot has drained, and a stand-in that waits for the kernel it is meant to run beside rehearses nothing)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(32))) rehearsal_exchange_kernel(RehearsalPieces pc) {
  for (int i = 0; i < pc.count; ++i) copy_words(pc.src[i], pc.dst[i], pc.n[i], (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// The harsher stand-in (FFX_DEBUG_VARIANT=rehearsal_paced[=GB/s per link]): what RCCL's send/recv kernel looks like to
// its neighbours -- ONE PERSISTENT workgroup per peer and direction (2 (N - 1) workgroups of 256 threads), each moving
// its piece in 64 KiB chunks and, between chunks, SPINNING on a flag in memory until the chunk "would have arrived" over
// a link of the given rate (default 50 GB/s, a third of an xGMI link's peak).  The workgroups therefore hold their CU
// slots and keep issuing polls for as long as the fabric would take, next to whatever the compute stream runs.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(32))) rehearsal_exchange_paced_kernel(RehearsalPieces pc) {
  const int i = blockIdx.x;
  if (i >= pc.count) return;
  const uint32_t* __restrict__ s = pc.src[i];
  uint32_t* __restrict__ d = pc.dst[i];
  const long n = pc.n[i];
  constexpr long kChunk = 16384;  // words = 64 KiB
  const long long t0 = wall_clock64();  // 100 MHz
  long done = 0;
  while (done < n) {
    const long m = n - done < kChunk ? n - done : kChunk;
    copy_words(s + done, d ? d + done : nullptr, m, threadIdx.x, blockDim.x);
    done += m;
    const long long due = t0 + (long long)((float)(done >> 8) * pc.ns_per_kib * 0.1f);  // ticks of 10 ns
    unsigned int seen = 0;
    while (wall_clock64() < due) seen += __builtin_nontemporal_load(pc.flag);  // poll memory, as a receiver waiting for its peer does
    if (seen == 0xFFFFFFFFu && threadIdx.x == 0 && d) d[0] = s[0];  // (never: keeps the polls alive)
  }
}

static int rehearsal_exchange(ffx_comm* c, int lane, const Piece* pc, hipStream_t st) {
  int64_t n_send = 0, n_recv = 0;
  C_TRY(check_pieces(c, pc, &n_send, &n_recv));
  const size_t need = (size_t)std::max<int64_t>(std::max(n_send, n_recv), 1);
  if (need > c->reh_cap) {
    C_HIP(hipStreamSynchronize(st));
    if (c->reh_src) (void)hipFree(c->reh_src);
    c->reh_src = nullptr;
    c->reh_cap = 0;
    const size_t want = need + need / 8;
    C_HIP(hipMalloc(reinterpret_cast<void**>(&c->reh_src), want * sizeof(float)));
    C_HIP(hipMemset(c->reh_src, 0, want * sizeof(float)));
    c->reh_cap = want;
  }
  RehearsalPieces k;
  k.count = 0;
  int64_t r_at = 0;
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) {  // exactly what the RCCL path does with the piece a rank keeps
      if (pc[p].n_send > 0 && pc[p].recv != pc[p].send)
        C_HIP(hipMemcpyAsync(pc[p].recv, pc[p].send, (size_t)pc[p].n_send * 4, hipMemcpyDeviceToDevice, st));
      continue;
    }
    if (pc[p].n_send > 0) {  // what peer p would read out of this GPU
      k.src[k.count] = static_cast<const uint32_t*>(pc[p].send);
      k.dst[k.count] = nullptr;  // read, not stored: the peer's memory is not on this GPU
      k.n[k.count++] = (long)pc[p].n_send;
    }
    if (pc[p].n_recv > 0) {  // what peer p would write into this GPU: out of the inbox (routing lane), else zeros
      const uint32_t* from = reinterpret_cast<const uint32_t*>(c->reh_src) + r_at;
      if (lane == 1 && c->inbox && c->inbox_at + (size_t)pc[p].n_recv <= c->inbox_words) {
        from = c->inbox + c->inbox_at;
        c->inbox_at += (size_t)pc[p].n_recv;
      }
      k.src[k.count] = from;
      k.dst[k.count] = static_cast<uint32_t*>(pc[p].recv);
      k.n[k.count++] = (long)pc[p].n_recv;
      r_at += pc[p].n_recv;
    }
  }
  if (k.count > 0) {
    long long gbs = 50;
    if (ffx_debug_variant("rehearsal_paced", &gbs)) {
      // a large piece travels over several channels of its link: sub-pieces of <= 4 MiB, a persistent workgroup each, the
      // link's rate shared between them
      RehearsalPieces cut;
      cut.count = 0;
      constexpr long kSub = 1 << 20;  // words
      int most = 1;
      for (int i = 0; i < k.count; ++i) most = std::max<int>(most, (int)((k.n[i] + kSub - 1) / kSub));
      most = std::min(most, std::max(1, (2 * kMaxWorld) / k.count));
      for (int i = 0; i < k.count; ++i) {
        const int parts = (int)std::min<long>(most, std::max<long>(1, (k.n[i] + kSub - 1) / kSub));
        for (int j = 0; j < parts; ++j) {
          const long lo = k.n[i] * j / parts, hi = k.n[i] * (j + 1) / parts;
          cut.src[cut.count] = k.src[i] + lo;
          cut.dst[cut.count] = k.dst[i] ? k.dst[i] + lo : nullptr;
          cut.n[cut.count++] = hi - lo;
        }
      }
      cut.ns_per_kib = 1024.0f / (float)std::max<long long>(gbs, 1) * (float)most;  // 1 KiB at g GB/s = 1024 / g ns; `most` channels share the link
      cut.flag = reinterpret_cast<const uint32_t*>(c->reh_src);
      k = cut;
      rehearsal_exchange_paced_kernel<<<k.count, 256, 0, st>>>(k);
    } else {
      k.ns_per_kib = 0.0f;
      k.flag = nullptr;
      rehearsal_exchange_kernel<<<16, 256, 0, st>>>(k);
    }
    C_HIP(hipGetLastError());
  }
  return FFX_OK;
}

// The routing lane, on first use (collective): a stream of its own and, over RCCL with peers, a second communicator whose
// id travels from rank 0 to the others through the first one.
static int comm_route_lane(ffx_comm* c) {
  if (c->route_stream || c->device < 0) return FFX_OK;
  C_HIP(hipSetDevice(c->device));
  hipStream_t st = nullptr;
  if (ffx_experiment("route_high_priority", nullptr)) {
    int least = 0, greatest = 0;
    C_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    C_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest));
  } else {
    C_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  }
  if (c->transport == FFX_COMM_RCCL && c->world > 1) {
    Rccl* r = rccl();
    ncclUniqueId uid;
    memset(&uid, 0, sizeof uid);
    int rc = FFX_OK;
    if (c->rank == 0 && r->GetUniqueId(&uid) != ncclSuccess) rc = ffx_fail(FFX_RCCL_ERROR, "ncclGetUniqueId failed for the routing lane");
    std::vector<char> all(sizeof uid * (size_t)c->world);
    if (rc == FFX_OK) rc = rccl_allgather_small(c, &uid, sizeof uid, all.data());
    if (rc == FFX_OK) {
      memcpy(&uid, all.data(), sizeof uid);  // rank 0's
      const ncclResult_t nr = r->CommInitRank(&c->nccl_route, c->world, uid, c->rank);
      if (nr != ncclSuccess) {
        c->nccl_route = nullptr;
        rc = ffx_fail(FFX_RCCL_ERROR, "ncclCommInitRank for the routing lane (rank %d of %d): %s", c->rank, c->world, r->GetErrorString(nr));
      }
    }
    if (rc != FFX_OK) {
      (void)hipStreamDestroy(st);
      return rc;
    }
  }
  c->route_stream = st;
  return FFX_OK;
}

// all-to-all-v of 4-byte words between DEVICE buffers on lane 0 / 1.  RCCL: asynchronous on `st`; host transport: blocks
// until the words have arrived.
static int comm_alltoallv(ffx_comm* c, int lane, const Piece* pc, hipStream_t st) {
  if (c->device < 0) return ffx_fail(FFX_BAD_ARG, "this communicator has no device: use ffx_comm_exchange_host");
  C_HIP(hipSetDevice(c->device));
  if (c->transport == FFX_COMM_HOST) return shm_exchange(c, pc, true, st);
  if (c->transport == FFX_COMM_REHEARSAL) return rehearsal_exchange(c, lane, pc, st);
  int64_t n_send = 0, n_recv = 0;
  C_TRY(check_pieces(c, pc, &n_send, &n_recv));
  // full mesh: one send and one receive per peer inside one group; the piece a rank keeps is a
  // device copy.  Zero-length pieces are skipped on both sides (the counts are symmetric by
  // construction: n_send for peer j here == n_recv for this rank on j).
  Rccl* r = rccl();
  ncclComm_t nc = lane == 0 ? c->nccl : c->nccl_route;
  bool grouped = false;
  for (int p = 0; p < c->world; ++p) {
    if (p == c->rank) {
      if (pc[p].n_send > 0 && pc[p].recv != pc[p].send)
        C_HIP(hipMemcpyAsync(pc[p].recv, pc[p].send, (size_t)pc[p].n_send * 4, hipMemcpyDeviceToDevice, st));
    } else if (pc[p].n_send > 0 || pc[p].n_recv > 0) {
      if (!nc) return ffx_fail(FFX_RCCL_ERROR, "lane %d of the communicator has not been brought up", lane);
      if (!grouped) {
        C_NCCL(r->GroupStart());
        grouped = true;
      }
      // (an error inside the group still closes it: a group left open would swallow the next collective of this thread)
      ncclResult_t nr = ncclSuccess;
      if (pc[p].n_send > 0) nr = r->Send(pc[p].send, (size_t)pc[p].n_send, ncclInt32, p, nc, st);
      if (nr == ncclSuccess && pc[p].n_recv > 0) nr = r->Recv(pc[p].recv, (size_t)pc[p].n_recv, ncclInt32, p, nc, st);
      if (nr != ncclSuccess) {
        (void)r->GroupEnd();
        return ffx_fail(FFX_RCCL_ERROR, "ncclSend / ncclRecv with rank %d: %s", p, r->GetErrorString(nr));
      }
    }
  }
  if (grouped) C_NCCL(r->GroupEnd());
  return FFX_OK;
}

int ffx_comm_exchange_host(ffx_comm* c, const float* send, const int64_t* send_counts, float* recv, const int64_t* recv_counts) {
  if (!c) return ffx_fail(FFX_BAD_ARG, "NULL comm");
  C_TRY(check_counts(c, send_counts, recv_counts));
  Piece pc[kMaxWorld];
  if (c->transport == FFX_COMM_HOST) {
    pieces_of(c, send, send_counts, recv, recv_counts, pc);
    return shm_exchange(c, pc, false, nullptr);
  }
  // RCCL: the same all-to-all-v, staged through device scratch kept with the communicator (a streamed index
  // produces its local scores on the host: they are PCIe-bound already)
  int64_t n_send = 0, n_recv = 0;
  for (int r = 0; r < c->world; ++r) {
    if (send_counts[r] < 0 || recv_counts[r] < 0) return ffx_fail(FFX_BAD_ARG, "negative count for peer %d", r);
    n_send += send_counts[r];
    n_recv += recv_counts[r];
  }
  C_HIP(hipSetDevice(c->device));
  const size_t need = (size_t)std::max<int64_t>(n_send + n_recv, 1);
  if (need > c->stage_cap) {
    if (c->stage_dev) (void)hipFree(c->stage_dev);
    c->stage_dev = nullptr;
    c->stage_cap = 0;
    C_HIP(hipMalloc(reinterpret_cast<void**>(&c->stage_dev), (need + need / 4) * sizeof(float)));
    c->stage_cap = need + need / 4;
  }
  float* dev = c->stage_dev;
  int rc = FFX_OK;
  if (n_send > 0 && ffx_h2d(c->device, dev, send, (size_t)n_send * sizeof(float), c->stream) != FFX_OK)
    rc = ffx_fail(FFX_HIP_ERROR, "upload of the local scores failed");
  if (rc == FFX_OK) rc = ffx_comm_exchange(c, dev, send_counts, dev + n_send, recv_counts, c->stream);
  if (rc == FFX_OK && n_recv > 0 && ffx_d2h(c->device, recv, dev + n_send, (size_t)n_recv * sizeof(float), c->stream) != FFX_OK)
    rc = ffx_fail(FFX_HIP_ERROR, "download of the received scores failed");
  if (hipStreamSynchronize(c->stream) != hipSuccess && rc == FFX_OK) rc = ffx_fail(FFX_HIP_ERROR, "stream synchronisation failed");
  return rc;
}

int ffx_comm_exchange(ffx_comm* c, const float* send_dev, const int64_t* send_counts, float* recv_dev, const int64_t* recv_counts,
                      void* stream) {
  if (!c) return ffx_fail(FFX_BAD_ARG, "NULL comm");
  if (c->device < 0) return ffx_fail(FFX_BAD_ARG, "this communicator has no device: use ffx_comm_exchange_host");
  C_TRY(check_counts(c, send_counts, recv_counts));
  Piece pc[kMaxWorld];
  pieces_of(c, send_dev, send_counts, recv_dev, recv_counts, pc);
  return comm_alltoallv(c, 0, pc, stream ? (hipStream_t)stream : c->stream);
}

// rehearsal only: what the peers would send on the routing lane from now on (n_words 4-byte words of DEVICE memory the
// caller keeps alive; NULL: zeros again).  The exchanges of that lane take their received pieces from it, front to back.
int ffx_comm_rehearsal_inbox(ffx_comm* c, const void* dev_words, int64_t n_words) {
  if (!c || n_words < 0) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (c->transport != FFX_COMM_REHEARSAL) return ffx_fail(FFX_BAD_ARG, "only the rehearsal transport has an inbox");
  c->inbox = static_cast<const uint32_t*>(dev_words);
  c->inbox_words = dev_words ? (size_t)n_words : 0;
  c->inbox_at = 0;
  return FFX_OK;
}

// ---- routing: which rank scores which candidate, and where each score lands at the owner ----------
int ffx_query_partition(int64_t n_queries, int world, int64_t* bounds) {
  if (!bounds || world < 1 || n_queries < 0) return ffx_fail(FFX_BAD_ARG, "bad argument");
  for (int r = 0; r <= world; ++r) bounds[r] = (int64_t)(((__int128)r * n_queries) / world);
  return FFX_OK;
}

int ffx_route(const int64_t* pair_off, int64_t n_queries, const int32_t* pair_ids, const int32_t* holder,
              const int64_t* doc_bounds, int rank, int world, int64_t* local_pair_off, int32_t* local_pair_ids,
              int64_t* send_counts, int64_t* recv_counts, int32_t* recv_slot, int64_t* own, int64_t* bad_pair) {
  if (bad_pair) *bad_pair = -1;
  if (!pair_off || n_queries < 0 || world < 1 || world > kMaxWorld || rank < 0 || rank >= world || !local_pair_off || !send_counts || !recv_counts || !own)
    return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (!holder && !doc_bounds) return ffx_fail(FFX_BAD_ARG, "either holder[] or doc_bounds[] must be given");
  if (pair_off[0] != 0) return ffx_fail(FFX_BAD_ARG, "pair_off[0] must be 0");
  for (int64_t q = 0; q < n_queries; ++q)
    if (pair_off[q + 1] < pair_off[q]) return ffx_fail(FFX_BAD_ARG, "pair_off not monotone at query %lld", (long long)q);
  const int64_t P = pair_off[n_queries];
  if (P > 0 && !pair_ids) return ffx_fail(FFX_BAD_ARG, "pair_ids is NULL");
  if (P > INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "more than 2^31-2 pairs in one batch: split the queries");
  const int64_t base = doc_bounds ? doc_bounds[rank] : 0;
  std::vector<int64_t> qb((size_t)world + 1);
  ffx_query_partition(n_queries, world, qb.data());
  const int64_t q_lo = qb[rank], q_hi = qb[rank + 1], lo = pair_off[q_lo], hi = pair_off[q_hi];
  own[0] = q_lo;
  own[1] = q_hi;
  own[2] = lo;
  own[3] = hi - lo;
  // The holder of every pair is worked out ONCE (threads over blocks of queries) and kept as a byte; the passes
  // below -- this rank's share of every query, the local CSR, and the owner side's arrival slots -- read it.
  int n_thr = 1;
  {
    cpu_set_t set;
    CPU_ZERO(&set);
    int cores = (int)std::max(1u, std::thread::hardware_concurrency());
    if (sched_getaffinity(0, sizeof set, &set) == 0) cores = std::min(cores, std::max(1, CPU_COUNT(&set)));
    n_thr = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)32, (int64_t)cores, P / (1 << 18) + 1, std::max<int64_t>(n_queries, 1)}));
  }
  std::vector<uint8_t> who;
  try {
    who.resize((size_t)std::max<int64_t>(P, 1));
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  }
  std::atomic<int64_t> first_bad(INT64_MAX);
  auto in_threads = [&](auto&& body) {  // body(t) for t in 0..n_thr-1; started threads are joined whatever happens
    std::vector<std::thread> pool;
    pool.reserve((size_t)n_thr);
    try {
      for (int t = 1; t < n_thr; ++t) pool.emplace_back([&body, t] { body(t); });
    } catch (...) {
      for (auto& th : pool) th.join();
      throw;
    }
    body(0);
    for (auto& th : pool) th.join();
  };
  try {
    // pass 1: holders, and how many pairs of each query this rank holds
    local_pair_off[0] = 0;
    in_threads([&](int t) {
      int64_t bad = INT64_MAX;
      for (int64_t q = n_queries * t / n_thr; q < n_queries * (t + 1) / n_thr; ++q) {
        int64_t mine = 0;
        for (int64_t p = pair_off[q]; p < pair_off[q + 1]; ++p) {
          int h;
          if (holder) {
            h = holder[p];
          } else {
            const int64_t id = pair_ids[p];
            h = (id < doc_bounds[0] || id >= doc_bounds[world]) ? -1 : (int)(std::upper_bound(doc_bounds, doc_bounds + world + 1, id) - doc_bounds) - 1;
          }
          if (h < 0 || h >= world) {
            if (bad == INT64_MAX) bad = p;
            h = 255;
          }
          who[(size_t)p] = (uint8_t)h;
          mine += h == rank;
        }
        local_pair_off[q + 1] = mine;  // counts now, offsets after the scan below
      }
      int64_t cur = first_bad.load();
      while (bad < cur && !first_bad.compare_exchange_weak(cur, bad)) {
      }
    });
    if (first_bad.load() != INT64_MAX) {
      if (bad_pair) *bad_pair = first_bad.load();
      return ffx_fail(FFX_NOT_FOUND, "pair %lld names a document no rank of the sharded index holds", (long long)first_bad.load());
    }
    for (int64_t q = 0; q < n_queries; ++q) local_pair_off[q + 1] += local_pair_off[q];
    for (int r = 0; r < world; ++r) send_counts[r] = local_pair_off[qb[r + 1]] - local_pair_off[qb[r]];
    // pass 2: the local CSR (ordinals relative to the shard)
    if (local_pair_ids)
      in_threads([&](int t) {
        for (int64_t q = n_queries * t / n_thr; q < n_queries * (t + 1) / n_thr; ++q) {
          int64_t at = local_pair_off[q];
          for (int64_t p = pair_off[q]; p < pair_off[q + 1]; ++p)
            if (who[(size_t)p] == rank) local_pair_ids[at++] = (int32_t)(pair_ids[p] - (holder ? 0 : base));
        }
      });
    // owner side: candidates of the owned queries arrive grouped by source rank, first-stage order inside.  Blocks of
    // the owned pairs are counted per source by threads; a block's first slot for a source follows from the counts
    // of the blocks before it.
    const int64_t n_own = hi - lo;
    std::vector<int64_t> block_count((size_t)n_thr * kMaxWorld, 0);
    in_threads([&](int t) {
      int64_t* c = block_count.data() + (size_t)t * kMaxWorld;
      for (int64_t p = lo + n_own * t / n_thr; p < lo + n_own * (t + 1) / n_thr; ++p) ++c[who[(size_t)p]];
    });
    int64_t acc = 0;
    for (int r = 0; r < world; ++r) {
      int64_t total = 0;
      for (int t = 0; t < n_thr; ++t) {
        const int64_t c = block_count[(size_t)t * kMaxWorld + r];
        block_count[(size_t)t * kMaxWorld + r] = acc + total;  // becomes: first slot of block t for source r
        total += c;
      }
      recv_counts[r] = total;
      acc += total;
    }
    if (recv_slot)
      in_threads([&](int t) {
        int64_t next[kMaxWorld];
        for (int r = 0; r < world; ++r) next[r] = block_count[(size_t)t * kMaxWorld + r];
        for (int64_t p = lo + n_own * t / n_thr; p < lo + n_own * (t + 1) / n_thr; ++p) recv_slot[p - lo] = (int32_t)next[who[(size_t)p]]++;
      });
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  } catch (const std::system_error&) {
    return ffx_fail(FFX_OOM, "could not start host threads");
  }
  return FFX_OK;
}

}  // extern "C"

// ---- one rank of a row-sharded index: local gather_dot + exchange + owner merge ------------------------
// A batch lives in a SLOT: the plan that scores this rank's share of it, the merge that holds the owner-side state of
// the queries this rank owns, and the buffer their scores arrive in.  Three slots rotate when every batch is new
// (ffx_sharded_stage_owned): while batch i is scored, batch i-1 is exchanged and then merged BEHIND batch i's
// gather_dot, and batch i+1 is routed into the third slot.  A batch that is set once and re-run (ffx_sharded_set_pairs)
// sits in slots 0 and 1, which alternate as send buffers and share one merge.
constexpr int kSlots = 3;

struct ffx_sharded {
  ffx_index* shard = nullptr;
  ffx_comm* comm = nullptr;
  int device = 0;
  std::vector<int64_t> doc_bounds;
  ffx_plan* plans[kSlots] = {nullptr, nullptr, nullptr};
  ffx_merge* merges[kSlots] = {nullptr, nullptr, nullptr};
  hipStream_t compute = nullptr;
  hipStream_t out = nullptr;  // downloads of finished batches (ffx_sharded_fetch_async), created on first use
  hipStream_t up = nullptr;   // uploads of a fresh batch's first-stage scores and query vectors, created on first use
  hipEvent_t ids_up = nullptr;  // the ordinals of the batch being staged have gone up (the link is free for the rest)
  hipEvent_t x_begin = nullptr, x_end = nullptr, m_begin = nullptr, m_end = nullptr;  // timing of the last step's exchange and merge
  hipEvent_t r_begin = nullptr, r_end = nullptr;                                      // ... and of the last batch's routing
  struct Slot {
    // scores arriving for the owned queries: one buffer per slot, because the exchange of batch i runs (on the
    // communicator's stream) while batch i-1 is still being merged out of another one (on the compute stream)
    void* recv = nullptr;
    size_t recv_cap = 0;
    std::vector<int64_t> send_counts, recv_counts;  // of the score exchange
    int64_t own[4] = {0, 0, 0, 0};                  // first owned query, one past the last, first owned pair, owned pairs
    int64_t local_pairs = 0;
    bool have_sparse = false;
    float* send_dev = nullptr;
    int merge = 0;  // the merge that holds this batch's owner-side state
    // ordering between the streams (no timing): the batch is routed / scored / its scores have left / it is merged
    hipEvent_t routed = nullptr, scored = nullptr, sent = nullptr, merged = nullptr;
    hipEvent_t queries_up = nullptr, sparse_up = nullptr;  // this rank's own query vectors / first-stage scores have been uploaded
    bool routed_valid = false, scored_valid = false, sent_valid = false, merged_valid = false;
  } slot[kSlots];
  hipEvent_t out_done[kSlots] = {nullptr, nullptr, nullptr};  // per MERGE: its results have been copied out
  bool out_valid[kSlots] = {false, false, false};
  // the batch whose exchange has been issued but whose owner-side merge has not (-1: none), with its arguments
  int pending = -1;
  double pending_alpha = 0.0;
  uint32_t pending_flags = 0;
  int last_merged = -1;  // slot whose merge was queued last: what ffx_sharded_fetch[_async] hands out
  bool fresh = false;    // the current batch came through ffx_sharded_stage_owned
  int cur = 0;           // ... into this slot
  uint64_t staged = 0;
  // what the set-once form fills (then published to slots 0 and 1)
  std::vector<int64_t> send_counts, recv_counts;
  // device-side routing: the batch's ids are uploaded once and everything else is derived
  // in HBM; buffers grow as needed and are kept
  struct DevArr {
    void* p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
      if (bytes <= cap) return FFX_OK;
      if (p) (void)hipFree(p);
      p = nullptr;
      cap = 0;
      const size_t want = bytes + bytes / 8 + 256;
      if (hipMalloc(&p, want) != hipSuccess) {
        (void)hipGetLastError();
        p = nullptr;
        return ffx_fail(FFX_OOM, "hipMalloc(%zu bytes) failed", want);
      }
      cap = want;
      return FFX_OK;
    }
    void release() {
      if (p) (void)hipFree(p);
      p = nullptr;
      cap = 0;
    }
    template <class T>
    T* as() const { return static_cast<T*>(p); }
  };
  DevArr d_ids, d_off, d_who, d_cnt, d_local_off, d_local_ids, d_pos, d_keys, d_arrival, d_slot, d_small, d_tmp;
  DevArr d_held, d_start, d_sendids, d_qcnt, d_verdict, d_cntq, d_tiles;  // owner-side routing (ffx_sharded_stage_owned)
  DevArr d_code_table;     // ffx_sharded_set_code_table: this rank's ordinal of every code (-1: not held here)
  int64_t n_codes = -1;    // -1: candidates come as ordinals
  std::vector<int32_t> tile_first;
  std::vector<int64_t> tile_p0, tile_hi;
  void* host_small = nullptr;                            // page-locked read-back block of the staging
  size_t host_small_cap = 0;
  // routing outputs, kept between calls (a fresh 140 MB array costs more in page faults than the routing itself)
  std::vector<int64_t> local_off;
  std::vector<int32_t> local_ids, slot_host;
  int64_t own[4] = {0, 0, 0, 0};
  int64_t local_pairs = 0;
  bool have_pairs = false, have_sparse = false;
  uint64_t runs = 0;
  int last = 0;
  // per-batch intake of the owner-routed form, for the stats
  int64_t stage_h2d_bytes = 0, stage_route_send_bytes = 0, stage_route_recv_bytes = 0;
  int64_t staged_query_bytes = 0;  // of ffx_sharded_stage_queries for the batch being staged
  double stage_host_ms = 0.0;
};

namespace {

// room in a slot for the scores its owned queries receive
int slot_recv_room(ffx_sharded* s, int i, int64_t own_pairs) {
  ffx_sharded::Slot& sl = s->slot[i];
  const size_t need = (size_t)std::max<int64_t>(own_pairs, 1) * sizeof(float);
  if (need <= sl.recv_cap) return FFX_OK;
  C_HIP(hipSetDevice(s->device));
  if (sl.recv) C_HIP(hipFree(sl.recv));
  sl.recv = nullptr;
  sl.recv_cap = 0;
  C_HIP(hipMalloc(&sl.recv, need + need / 8));
  sl.recv_cap = need + need / 8;
  return FFX_OK;
}

// what both routes of the set-once form end with: the batch sits in slots 0 and 1 (two plans alternating as send
// buffers, one merge), with room for the scores the owned queries receive
int sharded_finish_set(ffx_sharded* s, bool have_sparse) {
  s->have_sparse = have_sparse;
  for (int i = 0; i < 2; ++i) {
    ffx_sharded::Slot& sl = s->slot[i];
    C_TRY(slot_recv_room(s, i, s->own[3]));
    sl.send_counts = s->send_counts;
    sl.recv_counts = s->recv_counts;
    for (int k = 0; k < 4; ++k) sl.own[k] = s->own[k];
    sl.local_pairs = s->local_pairs;
    sl.have_sparse = have_sparse;
    sl.merge = 0;
    sl.routed_valid = sl.scored_valid = sl.sent_valid = sl.merged_valid = false;  // (sharded_set drained the streams)
  }
  for (bool& v : s->out_valid) v = false;
  s->fresh = false;
  s->last_merged = -1;
  s->have_pairs = true;
  return FFX_OK;
}

// ---- routing on the device (document-range shards) ----------------------------------------------------------------
// What ffx_route computes on the host, from ONE upload of the batch's ids:
//   mark     (a workgroup per query) holder of every pair as a byte, this rank's share of the query, and -- inside the
//            block of queries this rank owns -- how many scores every source rank will send
//   scan     shares -> offsets of the local CSR
//   compact  (a workgroup per query) the pairs this rank holds, in first-stage order, as shard-relative ordinals
//   arrival  the owned pairs sorted by holder with a stable radix pass over the holder byte: position i of the sorted
//            list is where that pair's score arrives (sources in rank order, first-stage order inside), so
//            recv_slot[pair] = i
struct RouteBounds {
  int64_t b[kMaxWorld + 1];
};

__global__ void __launch_bounds__(256)
route_mark_kernel(const int64_t* __restrict__ pair_off, const int32_t* __restrict__ ids, RouteBounds bounds, int world, int rank, int64_t own_lo,
                  int64_t own_hi, uint8_t* __restrict__ who, int64_t* __restrict__ share, unsigned long long* __restrict__ recv_counts,
                  unsigned long long* __restrict__ bad) {
  __shared__ unsigned int hist[kMaxWorld];
  __shared__ int part[4];
  const int q = blockIdx.x;
  if (threadIdx.x < kMaxWorld) hist[threadIdx.x] = 0;
  __syncthreads();
  int mine = 0;
  for (int64_t p = pair_off[q] + threadIdx.x; p < pair_off[q + 1]; p += blockDim.x) {
    const int64_t id = ids[p];
    int h = 255;
    if (id >= bounds.b[0] && id < bounds.b[world]) {
      int lo = 0, hi = world;  // the h with b[h] <= id < b[h+1] (empty ranges are skipped by construction)
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (bounds.b[mid] <= id) lo = mid; else hi = mid;
      }
      h = lo;
    } else {
      atomicMin(bad, (unsigned long long)p);
    }
    who[p] = (uint8_t)h;
    mine += h == rank;
    if (h != 255 && p >= own_lo && p < own_hi) atomicAdd(&hist[h], 1u);
  }
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
  __syncthreads();
  if (threadIdx.x == 0) share[q] = (int64_t)part[0] + part[1] + part[2] + part[3];
  if (threadIdx.x < world && hist[threadIdx.x]) atomicAdd(&recv_counts[threadIdx.x], (unsigned long long)hist[threadIdx.x]);
}

__global__ void __launch_bounds__(256)
route_compact_kernel(const int64_t* __restrict__ pair_off, const int32_t* __restrict__ ids, const uint8_t* __restrict__ who, int rank, int64_t base,
                     const int64_t* __restrict__ local_off, int32_t* __restrict__ local_ids) {
  __shared__ int wave_total[4];
  const int q = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int64_t at = local_off[q];
  const int64_t lo = pair_off[q], hi = pair_off[q + 1];
  for (int64_t p0 = lo; p0 < hi; p0 += blockDim.x) {
    const int64_t p = p0 + threadIdx.x;
    const bool keep = p < hi && who[p] == rank;
    const unsigned long long ballot = __ballot(keep);
    const int before = __popcll(ballot & ((1ull << lane) - 1ull));
    if (lane == 0) wave_total[wave] = __popcll(ballot);
    __syncthreads();
    int wave_base = 0, chunk = 0;
    for (int w = 0; w < 4; ++w) {
      if (w < wave) wave_base += wave_total[w];
      chunk += wave_total[w];
    }
    if (keep) local_ids[at + wave_base + before] = (int32_t)(ids[p] - base);
    at += chunk;
    __syncthreads();
  }
}

__global__ void __launch_bounds__(256) route_iota_kernel(int32_t* __restrict__ v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (int32_t)i;
}

__global__ void __launch_bounds__(256) route_invert_kernel(const int32_t* __restrict__ arrival, int64_t n, int32_t* __restrict__ slot) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) slot[arrival[i]] = (int32_t)i;
}

__global__ void route_pick_kernel(const int64_t* __restrict__ local_off, const int64_t* __restrict__ q_bounds, int world, int64_t n_queries,
                                  int64_t* __restrict__ out) {
  const int r = threadIdx.x;
  if (r <= world) out[r] = local_off[q_bounds[r]];
  if (r == 0) out[world + 1] = local_off[n_queries];
}

// The device form of sharded_set's routing.  On return: the two plans hold this rank's local CSR, the merge holds the
// owned queries' offsets / arrival slots / first-stage scores, send_counts / recv_counts / own are filled.
int sharded_route_on_device(ffx_sharded* s, const int64_t* pair_off, int64_t n_queries, const int32_t* pair_ids, const float* sparse, int mode,
                            int rank, int world, int64_t* bad_pair) {
  const int64_t P = pair_off[n_queries];
  std::vector<int64_t> qb((size_t)world + 1);
  ffx_query_partition(n_queries, world, qb.data());
  const int64_t q_lo = qb[(size_t)rank], q_hi = qb[(size_t)rank + 1], lo = pair_off[q_lo], hi = pair_off[q_hi], n_own = hi - lo;
  s->own[0] = q_lo;
  s->own[1] = q_hi;
  s->own[2] = lo;
  s->own[3] = n_own;
  C_HIP(hipSetDevice(s->device));
  hipStream_t st = s->compute;
  const size_t pb = (size_t)std::max<int64_t>(P, 1), ob = (size_t)std::max<int64_t>(n_own, 1);
  C_TRY(s->d_ids.ensure(pb * 4));
  C_TRY(s->d_off.ensure(((size_t)n_queries + 1) * 8));
  C_TRY(s->d_who.ensure(pb));
  C_TRY(s->d_cnt.ensure(((size_t)n_queries + 1) * 8));
  C_TRY(s->d_local_off.ensure(((size_t)n_queries + 1) * 8));
  C_TRY(s->d_local_ids.ensure(pb * 4));
  C_TRY(s->d_pos.ensure(ob * 4));
  C_TRY(s->d_keys.ensure(ob));
  C_TRY(s->d_arrival.ensure(ob * 4));
  C_TRY(s->d_slot.ensure(ob * 4));
  // small block: [0] bad pair, [1 .. 1+world) recv counts, then world+1 query bounds, then world+2 picked offsets
  const size_t small_words = 1 + (size_t)kMaxWorld + ((size_t)kMaxWorld + 1) + ((size_t)kMaxWorld + 2);
  C_TRY(s->d_small.ensure(small_words * 8));
  unsigned long long* d_bad = s->d_small.as<unsigned long long>();
  unsigned long long* d_recv = d_bad + 1;
  int64_t* d_qb = reinterpret_cast<int64_t*>(d_bad) + 1 + kMaxWorld;
  int64_t* d_pick = d_qb + kMaxWorld + 1;
  C_TRY(ffx_h2d(s->device, s->d_off.p, pair_off, ((size_t)n_queries + 1) * 8, st));
  if (P > 0) C_TRY(ffx_h2d(s->device, s->d_ids.p, pair_ids, (size_t)P * 4, st));
  C_TRY(ffx_h2d(s->device, d_qb, qb.data(), ((size_t)world + 1) * 8, st));
  C_HIP(hipMemsetAsync(d_bad, 0xFF, 8, st));
  C_HIP(hipMemsetAsync(d_recv, 0, (size_t)kMaxWorld * 8, st));
  C_HIP(hipMemsetAsync(s->d_cnt.p, 0, ((size_t)n_queries + 1) * 8, st));
  RouteBounds bounds;
  for (int r = 0; r <= world; ++r) bounds.b[r] = s->doc_bounds[(size_t)r];
  if (n_queries > 0) {
    route_mark_kernel<<<(unsigned)n_queries, 256, 0, st>>>(s->d_off.as<int64_t>(), s->d_ids.as<int32_t>(), bounds, world, rank, lo, hi, s->d_who.as<uint8_t>(),
                                                         s->d_cnt.as<int64_t>(), d_recv, d_bad);
    C_HIP(hipGetLastError());
  }
  {
    size_t tmp = 0;
    C_HIP(rocprim::exclusive_scan(nullptr, tmp, s->d_cnt.as<int64_t>(), s->d_local_off.as<int64_t>(), (int64_t)0, (size_t)n_queries + 1,
                                  rocprim::plus<int64_t>(), st));
    C_TRY(s->d_tmp.ensure(std::max<size_t>(tmp, 16)));
    C_HIP(rocprim::exclusive_scan(s->d_tmp.p, tmp, s->d_cnt.as<int64_t>(), s->d_local_off.as<int64_t>(), (int64_t)0, (size_t)n_queries + 1,
                                  rocprim::plus<int64_t>(), st));
  }
  if (n_queries > 0) {
    route_compact_kernel<<<(unsigned)n_queries, 256, 0, st>>>(s->d_off.as<int64_t>(), s->d_ids.as<int32_t>(), s->d_who.as<uint8_t>(), rank,
                                                            s->doc_bounds[(size_t)rank], s->d_local_off.as<int64_t>(), s->d_local_ids.as<int32_t>());
    C_HIP(hipGetLastError());
  }
  route_pick_kernel<<<1, kMaxWorld + 1, 0, st>>>(s->d_local_off.as<int64_t>(), d_qb, world, n_queries, d_pick);
  C_HIP(hipGetLastError());
  if (n_own > 0) {
    route_iota_kernel<<<(unsigned)((n_own + 255) / 256), 256, 0, st>>>(s->d_pos.as<int32_t>(), n_own);
    C_HIP(hipGetLastError());
    int bits = 1;
    while ((1 << bits) < world) ++bits;
    size_t tmp = 0;
    const uint8_t* keys_in = s->d_who.as<uint8_t>() + lo;
    C_HIP(rocprim::radix_sort_pairs(nullptr, tmp, keys_in, s->d_keys.as<uint8_t>(), s->d_pos.as<int32_t>(), s->d_arrival.as<int32_t>(), (size_t)n_own, 0, bits, st));
    C_TRY(s->d_tmp.ensure(std::max<size_t>(tmp, 16)));
    C_HIP(rocprim::radix_sort_pairs(s->d_tmp.p, tmp, keys_in, s->d_keys.as<uint8_t>(), s->d_pos.as<int32_t>(), s->d_arrival.as<int32_t>(), (size_t)n_own, 0, bits, st));
    route_invert_kernel<<<(unsigned)((n_own + 255) / 256), 256, 0, st>>>(s->d_arrival.as<int32_t>(), n_own, s->d_slot.as<int32_t>());
    C_HIP(hipGetLastError());
  }
  // what the host needs: the verdict, the counts, the local CSR offsets (the plans size their row lists from them)
  std::vector<int64_t> small(small_words);
  if (s->local_off.size() < (size_t)n_queries + 1) s->local_off.resize((size_t)n_queries + 1);
  C_TRY(ffx_d2h(s->device, small.data(), s->d_small.p, small_words * 8, st));
  C_TRY(ffx_d2h(s->device, s->local_off.data(), s->d_local_off.p, ((size_t)n_queries + 1) * 8, st));
  C_HIP(hipStreamSynchronize(st));
  const unsigned long long bad = (unsigned long long)small[0];
  if (bad != ~0ull) {
    if (bad_pair) *bad_pair = (int64_t)bad;
    return ffx_fail(FFX_NOT_FOUND, "pair %llu names a document no rank of the sharded index holds", bad);
  }
  const int64_t* pick = small.data() + 1 + kMaxWorld + (kMaxWorld + 1);
  for (int r = 0; r < world; ++r) {
    s->recv_counts[(size_t)r] = small[1 + (size_t)r];
    s->send_counts[(size_t)r] = pick[r + 1] - pick[r];
  }
  s->local_pairs = pick[world + 1];
  // plans: the first takes the lists device to device, the second copies the first
  int64_t bad_local = -1;
  int rc = ffx_plan_set_pairs_device(s->plans[0], s->d_local_off.as<int64_t>(), s->local_off.data(), n_queries, s->d_local_ids.as<int32_t>(), mode, &bad_local);
  if (rc == FFX_NOT_FOUND && bad_pair && bad_local >= 0) {  // local position -> position in the caller's list
    const int64_t q = std::upper_bound(s->local_off.begin(), s->local_off.begin() + n_queries + 1, bad_local) - s->local_off.begin() - 1;
    int64_t seen = s->local_off[(size_t)q];
    for (int64_t p = pair_off[q]; p < pair_off[q + 1]; ++p)
      if (pair_ids[p] >= s->doc_bounds[(size_t)rank] && pair_ids[p] < s->doc_bounds[(size_t)rank + 1] && seen++ == bad_local) {
        *bad_pair = p;
        break;
      }
  }
  C_TRY(rc);
  C_TRY(ffx_plan_copy_pairs(s->plans[1], s->plans[0]));
  for (int i = 0; i < 2; ++i) {
    void* p = nullptr;
    C_TRY(ffx_plan_device_ptr(s->plans[i], FFX_BUF_FF, &p, nullptr));
    s->slot[i].send_dev = static_cast<float*>(p);
  }
  std::vector<int64_t> own_off((size_t)(q_hi - q_lo) + 1);
  for (int64_t q = q_lo; q <= q_hi; ++q) own_off[(size_t)(q - q_lo)] = pair_off[q] - lo;
  C_TRY(ffx_merge_set_device(s->merges[0], own_off.data(), q_hi - q_lo, s->d_slot.as<int32_t>(), sparse ? sparse + lo : nullptr));
  return FFX_OK;
}

int sharded_set(ffx_sharded* s, const int64_t* pair_off, int64_t n_queries, const int32_t* pair_ids, const int32_t* holder,
                const float* sparse, int mode, int64_t* bad_pair) {
  if (bad_pair) *bad_pair = -1;
  if (!s || !pair_off || n_queries < 0) return ffx_fail(FFX_BAD_ARG, "bad argument");
  int rank = 0, world = 1;
  ffx_comm_rank(s->comm, &rank, &world);
  C_TRY(ffx_sharded_sync(s));
  s->have_pairs = false;
  const int64_t P = pair_off[n_queries];
  s->send_counts.assign((size_t)world, 0);
  s->recv_counts.assign((size_t)world, 0);
  if (!holder && !s->doc_bounds.empty() && !ffx_debug_variant("route_host", nullptr)) {
    // document-range shards: the batch is routed on the device from one upload of its ids
    if (pair_off[0] != 0) return ffx_fail(FFX_BAD_ARG, "pair_off[0] must be 0");
    for (int64_t q = 0; q < n_queries; ++q)
      if (pair_off[q + 1] < pair_off[q]) return ffx_fail(FFX_BAD_ARG, "pair_off not monotone at query %lld", (long long)q);
    if (P > 0 && !pair_ids) return ffx_fail(FFX_BAD_ARG, "pair_ids is NULL");
    if (P > INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "more than 2^31-2 pairs in one batch: split the queries");
    if (n_queries > INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "too many queries");
    C_TRY(sharded_route_on_device(s, pair_off, n_queries, pair_ids, sparse, mode, rank, world, bad_pair));
    return sharded_finish_set(s, sparse != nullptr);
  }
  // one routing pass: the outputs are sized for the worst case (every pair local; the owned block is known beforehand)
  std::vector<int64_t> qb((size_t)world + 1);
  ffx_query_partition(n_queries, world, qb.data());
  std::vector<int64_t>& local_off = s->local_off;
  std::vector<int32_t>& local_ids = s->local_ids;
  std::vector<int32_t>& slot = s->slot_host;
  try {
    if (local_off.size() < (size_t)n_queries + 1) local_off.resize((size_t)n_queries + 1);
    if (local_ids.size() < (size_t)std::max<int64_t>(P, 1)) local_ids.resize((size_t)std::max<int64_t>(P, 1));
    const int64_t n_own = pair_off[qb[(size_t)rank + 1]] - pair_off[qb[(size_t)rank]];
    if (slot.size() < (size_t)std::max<int64_t>(n_own, 1)) slot.resize((size_t)std::max<int64_t>(n_own, 1));
  } catch (const std::bad_alloc&) {
    return ffx_fail(FFX_OOM, "host allocation failed");
  }
  C_TRY(ffx_route(pair_off, n_queries, pair_ids, holder, holder ? nullptr : s->doc_bounds.data(), rank, world, local_off.data(), local_ids.data(),
                  s->send_counts.data(), s->recv_counts.data(), slot.data(), s->own, bad_pair));
  s->local_pairs = local_off[n_queries];
  for (int i = 0; i < 2; ++i) {
    int64_t bad_local = -1;
    // the second plan scores the same lists: it takes them over from the first on the device
    int rc = i == 0 ? ffx_plan_set_pairs(s->plans[0], local_off.data(), n_queries, local_ids.data(), nullptr, mode, &bad_local)
                    : ffx_plan_copy_pairs(s->plans[1], s->plans[0]);
    if (rc == FFX_NOT_FOUND && bad_pair && bad_local >= 0) {  // local position -> position in the caller's list
      const int64_t q = std::upper_bound(local_off.begin(), local_off.begin() + n_queries + 1, bad_local) - local_off.begin() - 1;  // (the vector may be longer: it is kept between calls)
      int64_t seen = local_off[q];
      for (int64_t p = pair_off[q]; p < pair_off[q + 1]; ++p) {
        const bool mine = holder ? holder[p] == rank : (pair_ids[p] >= s->doc_bounds[rank] && pair_ids[p] < s->doc_bounds[rank + 1]);
        if (mine && seen++ == bad_local) {
          *bad_pair = p;
          break;
        }
      }
    }
    C_TRY(rc);
    void* p = nullptr;
    C_TRY(ffx_plan_device_ptr(s->plans[i], FFX_BUF_FF, &p, nullptr));
    s->slot[i].send_dev = static_cast<float*>(p);
  }
  const int64_t q_lo = s->own[0], q_hi = s->own[1], lo = s->own[2];
  std::vector<int64_t> own_off((size_t)(q_hi - q_lo) + 1);
  for (int64_t q = q_lo; q <= q_hi; ++q) own_off[(size_t)(q - q_lo)] = pair_off[q] - lo;
  C_TRY(ffx_merge_set(s->merges[0], own_off.data(), q_hi - q_lo, slot.data(), sparse ? sparse + lo : nullptr));
  return sharded_finish_set(s, sparse != nullptr);
}

// ---- owner-side routing: a FRESH batch taken in at 1/world of the cost per rank ------------------------------------------
// The reference scores a new set of queries on every call (index/base.py:418-429, 445-459).  With one process per GPU
// nobody holds "the batch": rank r is handed only the block of queries it OWNS (their candidate lists and first-stage
// scores: ~P/world pairs), finds the holder of each of its candidates on the device, and the ids travel to the holders in
// a first all-to-all on the communicator's routing lane; the holders score them and the scores come back as before.
//   count    (a workgroup per owned query) how many of the query's candidates each rank holds: cnt[h][q]
//   scan     cnt (holder-major) -> start[h][q]: where the (h, q) run begins in the holder-grouped list = the ARRIVAL
//            ORDER of the scores (sources in rank order, queries ascending, first-stage order inside)
//   place    (a workgroup per owned query) stable multi-way partition of the query's candidates by holder:
//            slot[p] = arrival position of candidate p (what the owner-side merge reads), send_ids[slot[p]] = its
//            ordinal on the holder
//   exchange A  cnt[h][*] -> holder h: every rank learns how many candidates of EVERY query it holds (its local CSR offsets)
//   exchange B  send_ids runs -> the holders, received straight into the scoring plan's candidate list: the messages of
//            the owners in rank order ARE the local CSR (queries are owned in blocks, in rank order)
//   exchange C  two words per peer: what went wrong, so that every rank leaves the call with the same verdict
__device__ inline int route_holder_of(int64_t id, const RouteBounds& bounds, int world) {
  if (id < bounds.b[0] || id >= bounds.b[world]) return -1;
  int lo = 0, hi = world;  // the h with b[h] <= id < b[h+1] (empty ranges are skipped by construction)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (bounds.b[mid] <= id) lo = mid; else hi = mid;
  }
  return lo;
}

// A TILE is up to kTile = 1024 consecutive candidates of one owned query (tile_p0 / tile_hi: its first candidate and the
// end of its query, made on the host from the offsets): a workgroup per tile, four passes of 256 candidates whose loads are
// all issued before anything is ranked, one barrier -- thousands of independent workgroups with short dependency chains
// instead of one workgroup per query walking it in twenty dependent steps next to a kernel that fills every CU.
// The distinct holders among a wave's 64 candidates are taken one at a time (ballots): `before` = how many of the lanes
// below hold the same holder's candidate; the wave's count per holder lands in its own row of LDS.
constexpr int kTile = 1024, kTilePasses = kTile / 256;

__device__ inline int wave_rank_by_holder(int h, bool valid, int lane, int* wave_row) {
  int before = 0;
  unsigned long long todo = __ballot(valid);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int hh = __shfl(h, leader, 64);
    const unsigned long long same = __ballot(valid && h == hh);
    if (valid && h == hh) before = __popcll(same & ((1ull << lane) - 1ull));
    if (lane == leader) wave_row[hh] = __popcll(same);
    todo &= ~same;
  }
  return before;
}

__global__ void __launch_bounds__(256)
own_count_kernel(const int64_t* __restrict__ tile_p0, const int64_t* __restrict__ tile_hi, const int32_t* __restrict__ ids, const int32_t* __restrict__ held,
                 RouteBounds bounds, int world, int n_tiles, int32_t* __restrict__ cnt, unsigned long long* __restrict__ bad) {
  __shared__ int wave_cnt[kTilePasses * 4][kMaxWorld];
  const int t = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = 0; j < kTilePasses; ++j)
    for (int i = lane; i < world; i += 64) wave_cnt[j * 4 + wave][i] = 0;
  const int64_t p0 = tile_p0[t] + threadIdx.x, hi = tile_hi[t];
  int h[kTilePasses];
#pragma unroll
  for (int j = 0; j < kTilePasses; ++j) {
    const int64_t p = p0 + j * 256;
    h[j] = p < hi ? (held ? held[p] : ids[p]) : -1;  // (an ordinal for now: every load is in flight before the first is used)
  }
#pragma unroll
  for (int j = 0; j < kTilePasses; ++j) {
    const int64_t p = p0 + j * 256;
    if (p < hi) {
      if (!held) h[j] = route_holder_of(h[j], bounds, world);
      if (h[j] < 0 || h[j] >= world) atomicMin(bad, (unsigned long long)p);
    }
    (void)wave_rank_by_holder(h[j], p < hi && h[j] >= 0 && h[j] < world, lane, wave_cnt[j * 4 + wave]);
  }
  __syncthreads();
  if (threadIdx.x < world) {
    int total = 0;
    for (int r = 0; r < kTilePasses * 4; ++r) total += wave_cnt[r][threadIdx.x];
    cnt[(size_t)threadIdx.x * n_tiles + t] = total;
  }
}

__global__ void __launch_bounds__(256)
own_place_kernel(const int64_t* __restrict__ tile_p0, const int64_t* __restrict__ tile_hi, const int32_t* __restrict__ ids, const int32_t* __restrict__ held,
                 RouteBounds bounds, int world, int n_tiles, const int32_t* __restrict__ start, int32_t* __restrict__ slot, int32_t* __restrict__ send_ids) {
  __shared__ int wave_cnt[kTilePasses * 4][kMaxWorld];
  const int t = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = 0; j < kTilePasses; ++j)
    for (int i = lane; i < world; i += 64) wave_cnt[j * 4 + wave][i] = 0;
  const int64_t p0 = tile_p0[t] + threadIdx.x, hi = tile_hi[t];
  int32_t id[kTilePasses];
  int h[kTilePasses], before[kTilePasses];
#pragma unroll
  for (int j = 0; j < kTilePasses; ++j) {
    const int64_t p = p0 + j * 256;
    id[j] = p < hi ? ids[p] : 0;
    h[j] = p < hi && held ? held[p] : -1;
  }
#pragma unroll
  for (int j = 0; j < kTilePasses; ++j) {
    const int64_t p = p0 + j * 256;
    if (p < hi && !held) h[j] = route_holder_of(id[j], bounds, world);
    if (h[j] >= world) h[j] = -1;  // (an id no rank holds was reported by the count pass and gets no slot)
    before[j] = wave_rank_by_holder(h[j], p < hi && h[j] >= 0, lane, wave_cnt[j * 4 + wave]);
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kTilePasses; ++j) {
    const int64_t p = p0 + j * 256;
    if (p >= hi) continue;
    if (h[j] >= 0) {
      int at = start[(size_t)h[j] * n_tiles + t] + before[j];
      for (int r = 0; r < j * 4 + wave; ++r) at += wave_cnt[r][h[j]];
      slot[p] = at;
      send_ids[at] = held ? id[j] : (int32_t)((int64_t)id[j] - bounds.b[h[j]]);
    } else {
      slot[p] = 0;  // never read: the call fails
    }
  }
}

// per-query counts out of the tile scan: cnt_q[h][q] = start[h][first tile of q + 1] - start[h][first tile of q] (message A);
// small[1 + h] = how many of the owned candidates rank h holds
__global__ void __launch_bounds__(256)
own_qcount_kernel(const int32_t* __restrict__ start, const int32_t* __restrict__ tile_first, int world, int n_q, int n_tiles, int32_t* __restrict__ cnt_q,
                  long long* __restrict__ small) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long)world * n_q) {
    const int h = (int)(i / n_q), q = (int)(i % n_q);
    cnt_q[i] = start[(size_t)h * n_tiles + tile_first[q + 1]] - start[(size_t)h * n_tiles + tile_first[q]];
  }
  if (i < world) small[1 + i] = (long long)start[(size_t)(i + 1) * n_tiles] - (long long)start[(size_t)i * n_tiles];
}

// candidates that travelled as CODES (a numbering every rank agrees on: a ranking's id codes) become this rank's ordinals
__global__ void __launch_bounds__(256) codes_to_ordinals_kernel(int32_t* __restrict__ ids, int64_t n, const int32_t* __restrict__ table, int64_t n_codes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t c = ids[i];
  ids[i] = (uint32_t)c < (uint64_t)n_codes ? table[c] : -1;  // -1: the ordinal check that follows reports it
}

// exchange C's words, made on the device: own_bad = the smallest position of one of MY candidates that no rank holds,
// plan_bad = the smallest LOCAL position of an ordinal that names no row here (either ~0: none)
__global__ void stage_verdict_kernel(const unsigned long long* __restrict__ own_bad, const unsigned long long* __restrict__ plan_bad, const int64_t* __restrict__ local_off,
                                     RouteBounds qb, int world, int host_status, int32_t* __restrict__ out) {
  const int r = threadIdx.x;
  if (r >= world) return;
  const unsigned long long ob = *own_bad, pb = plan_bad ? *plan_bad : ~0ull;
  int32_t report = -1;
  if (pb != ~0ull) {
    const long long lo = local_off[qb.b[r]], hi = local_off[qb.b[r + 1]];
    if ((long long)pb >= lo && (long long)pb < hi) report = (int32_t)((long long)pb - lo);
  }
  out[2 * r] = report;
  out[2 * r + 1] = ob != ~0ull ? 1 : host_status ? 2 : pb != ~0ull ? 3 : 0;
}

// The streams beside the three that carry kernels (compute, the communicator's two lanes), created on first use: `out`, on
// which results leave, and `up`, on which a fresh batch's first-stage scores and query vectors arrive (DMA both ways, beside
// the kernels).  A HIP stream is assigned one of FOUR hardware queues per priority class, and two streams on one queue
// execute in issue order: compute, the two lanes and the default stream fill the normal class, so these two live in the
// high (results: not behind the next batch's kernels, as ffx_score's result stream) and the low class.
int sharded_io_streams(ffx_sharded* s) {
  if (s->out && s->up) return FFX_OK;
  C_HIP(hipSetDevice(s->device));
  int least = 0, greatest = 0;
  C_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  if (!s->out) C_HIP(hipStreamCreateWithPriority(&s->out, hipStreamNonBlocking, greatest));
  if (!s->up) C_HIP(hipStreamCreateWithPriority(&s->up, hipStreamNonBlocking, least));
  if (!s->ids_up) C_HIP(hipEventCreateWithFlags(&s->ids_up, hipEventDisableTiming));
  return FFX_OK;
}

struct WidenI32 {
  __device__ int64_t operator()(int32_t v) const { return (int64_t)v; }
};

// One batch staged into slot (staged % kSlots): see above.  Collective; every rank leaves with the same status.
int sharded_stage_owned(ffx_sharded* s, const int64_t* q_bounds, const float* own_queries, const int64_t* own_off, const int32_t* own_ids,
                        const int32_t* own_holder, const float* own_sparse, int mode, int32_t order_base, int64_t* bad_pair) {
  if (bad_pair) *bad_pair = -1;
  if (!s || !q_bounds || !own_off) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (mode < FFX_MODE_PASSAGE || mode > FFX_MODE_AVEP) return ffx_fail(FFX_BAD_ARG, "unknown mode %d", mode);
  int rank = 0, world = 1;
  ffx_comm_rank(s->comm, &rank, &world);
  if (!own_holder && s->doc_bounds.empty()) return ffx_fail(FFX_BAD_ARG, "created without doc_bounds: the candidates need explicit holders");
  if (q_bounds[0] != 0) return ffx_fail(FFX_BAD_ARG, "q_bounds[0] must be 0");
  for (int r = 0; r < world; ++r)
    if (q_bounds[r + 1] < q_bounds[r]) return ffx_fail(FFX_BAD_ARG, "q_bounds not monotone at rank %d", r);
  const int64_t Q = q_bounds[world], q_lo = q_bounds[rank], Qo = q_bounds[rank + 1] - q_lo;
  if (Q > INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "too many queries");
  if (own_off[0] != 0) return ffx_fail(FFX_BAD_ARG, "own_pair_off[0] must be 0");
  int64_t longest = 0;
  for (int64_t q = 0; q < Qo; ++q) {
    if (own_off[q + 1] < own_off[q]) return ffx_fail(FFX_BAD_ARG, "own_pair_off not monotone at query %lld", (long long)q);
    longest = std::max(longest, own_off[q + 1] - own_off[q]);
  }
  const int64_t Po = own_off[Qo];
  if (Po > INT32_MAX - 1) return ffx_fail(FFX_BAD_ARG, "more than 2^31-2 pairs in one batch: split the queries");
  if (Po > 0 && !own_ids) return ffx_fail(FFX_BAD_ARG, "own_pair_ids is NULL");
  if ((int64_t)world * Qo > INT32_MAX - 2) return ffx_fail(FFX_BAD_ARG, "too many owned queries for %d ranks", world);
  const auto t_begin = std::chrono::steady_clock::now();
  C_TRY(comm_route_lane(s->comm));  // collective on first use
  C_HIP(hipSetDevice(s->device));
  hipStream_t rs = s->comm->route_stream;
  if (!s->fresh) {  // coming from the set-once form: nothing of it may still be in flight
    C_TRY(ffx_sharded_sync(s));
    for (auto& sl : s->slot) sl.routed_valid = sl.scored_valid = sl.sent_valid = sl.merged_valid = false;
    for (bool& v : s->out_valid) v = false;
    s->last_merged = -1;
  }
  s->have_pairs = false;
  const int t = (int)(s->staged % kSlots);
  ffx_sharded::Slot& sl = s->slot[t];
  // the batch that used this slot three stages ago: scored, its scores sent, merged, its results copied out
  if (sl.scored_valid) C_HIP(hipStreamWaitEvent(rs, sl.scored, 0));
  if (sl.sent_valid) C_HIP(hipStreamWaitEvent(rs, sl.sent, 0));
  if (sl.merged_valid) C_HIP(hipStreamWaitEvent(rs, sl.merged, 0));
  if (s->out_valid[t]) C_HIP(hipStreamWaitEvent(rs, s->out_done[t], 0));
  if (s->pending == t) return ffx_fail(FFX_BAD_ARG, "three batches staged without a run in between");
  const int qdim = [&] {
    int32_t d = 0;
    ffx_index_query_dim(s->shard, &d);
    return (int)d;
  }();
  // tiles: up to kTile consecutive candidates of one query (what a workgroup of the count / place kernels takes)
  std::vector<int64_t>& tile_hi = s->tile_hi;
  std::vector<int64_t>& tile_p0 = s->tile_p0;
  std::vector<int32_t>& tile_first = s->tile_first;
  tile_hi.clear();
  tile_p0.clear();
  tile_first.resize((size_t)Qo + 1);
  for (int64_t q = 0; q < Qo; ++q) {
    tile_first[(size_t)q] = (int32_t)tile_p0.size();
    for (int64_t p = own_off[q]; p < own_off[q + 1]; p += kTile) {
      tile_p0.push_back(p);
      tile_hi.push_back(own_off[q + 1]);
    }
  }
  tile_first[(size_t)Qo] = (int32_t)tile_p0.size();
  const int64_t n_tiles = (int64_t)tile_p0.size();
  if ((int64_t)world * n_tiles > INT32_MAX - 2) return ffx_fail(FFX_BAD_ARG, "too many candidates for %d ranks in one batch: split the queries", world);
  // room: the owner-side state goes straight into the slot's merge, the local lists straight into its plan
  int64_t* m_off = nullptr;
  int32_t* m_slot = nullptr;
  float* m_sparse = nullptr;
  C_TRY(ffx_merge_adopt_begin(s->merges[t], Qo, Po, own_sparse != nullptr, &m_off, &m_slot, &m_sparse));
  float* d_q = nullptr;
  if (own_queries) C_TRY(ffx_plan_queries_room(s->plans[t], Q, &d_q));
  const size_t pb = (size_t)std::max<int64_t>(Po, 1), n_cnt = (size_t)world * (size_t)n_tiles + 1, tb = (size_t)std::max<int64_t>(n_tiles, 1);
  C_TRY(s->d_ids.ensure(pb * 4));
  if (own_holder) C_TRY(s->d_held.ensure(pb * 4));
  C_TRY(s->d_cnt.ensure(n_cnt * 4));
  C_TRY(s->d_start.ensure(n_cnt * 4));
  C_TRY(s->d_cntq.ensure(((size_t)world * (size_t)Qo + 1) * 4));
  C_TRY(s->d_tiles.ensure(tb * 16 + ((size_t)Qo + 1) * 4 + 64));
  C_TRY(s->d_sendids.ensure(pb * 4));
  C_TRY(s->d_qcnt.ensure(((size_t)Q + 1) * 4));
  C_TRY(s->d_local_off.ensure(((size_t)Q + 1) * 8));
  const size_t small_words = 1 + (size_t)kMaxWorld;
  C_TRY(s->d_small.ensure(small_words * 8));
  C_TRY(s->d_verdict.ensure(4 * (size_t)kMaxWorld * 4));
  const size_t back_bytes = small_words * 8 + ((size_t)Q + 1) * 8 + 4 * (size_t)kMaxWorld * 4;
  if (back_bytes > s->host_small_cap) {
    if (s->host_small) (void)hipHostFree(s->host_small);
    s->host_small = nullptr;
    s->host_small_cap = 0;
    C_HIP(hipHostMalloc(&s->host_small, back_bytes + back_bytes / 4, hipHostMallocDefault));
    s->host_small_cap = back_bytes + back_bytes / 4;
  }
  long long* h_small = static_cast<long long*>(s->host_small);
  int64_t* h_local_off = reinterpret_cast<int64_t*>(h_small + small_words);
  int32_t* h_verdict = reinterpret_cast<int32_t*>(h_local_off + Q + 1);
  unsigned long long* d_bad = s->d_small.as<unsigned long long>();
  int64_t* d_tile_p0 = s->d_tiles.as<int64_t>();
  int64_t* d_tile_hi = d_tile_p0 + tb;
  int32_t* d_tile_first = reinterpret_cast<int32_t*>(d_tile_hi + tb);
  // uploads: this rank's block only, one after the other on a stream of their own (the link is the shared resource) in the
  // order the routing stream needs them: the query vectors (their all-gather starts at once and runs while the ordinals are
  // still on their way), the ordinals (the routing kernels), the first-stage scores (nobody before the owner-side merge).
  C_TRY(sharded_io_streams(s));
  hipStream_t io = s->up;
  if (sl.scored_valid) C_HIP(hipStreamWaitEvent(io, sl.scored, 0));
  if (sl.merged_valid) C_HIP(hipStreamWaitEvent(io, sl.merged, 0));
  if (s->out_valid[t]) C_HIP(hipStreamWaitEvent(io, s->out_done[t], 0));
  C_HIP(hipEventRecord(s->r_begin, rs));
  int64_t h2d = (Qo + 1) * 8 + s->staged_query_bytes;
  s->staged_query_bytes = 0;
  if (own_queries && Qo > 0) {
    C_TRY(ffx_h2d(s->device, d_q + (size_t)q_lo * qdim, own_queries, (size_t)Qo * qdim * 4, io));
    h2d += Qo * (int64_t)qdim * 4;
  }
  C_HIP(hipEventRecord(sl.queries_up, io));
  if (Po > 0) {
    C_TRY(ffx_h2d(s->device, s->d_ids.p, own_ids, (size_t)Po * 4, io));
    h2d += Po * 4;
    if (own_holder) {
      C_TRY(ffx_h2d(s->device, s->d_held.p, own_holder, (size_t)Po * 4, io));
      h2d += Po * 4;
    }
  }
  C_HIP(hipEventRecord(s->ids_up, io));
  if (own_sparse && Po > 0) {
    C_TRY(ffx_h2d(s->device, m_sparse, own_sparse, (size_t)Po * 4, io));
    h2d += Po * 4;
  }
  C_HIP(hipEventRecord(sl.sparse_up, io));
  C_TRY(ffx_h2d(s->device, m_off, own_off, ((size_t)Qo + 1) * 8, rs));
  if (n_tiles > 0) {
    C_TRY(ffx_h2d(s->device, d_tile_p0, tile_p0.data(), (size_t)n_tiles * 8, rs));
    C_TRY(ffx_h2d(s->device, d_tile_hi, tile_hi.data(), (size_t)n_tiles * 8, rs));
  }
  C_TRY(ffx_h2d(s->device, d_tile_first, tile_first.data(), ((size_t)Qo + 1) * 4, rs));
  int64_t route_send = 0, route_recv = 0;
  // the query vectors of the whole batch, every rank contributing its block
  if (own_queries && world > 1) {
    Piece qp[kMaxWorld];
    for (int r = 0; r < world; ++r) {
      qp[r].send = d_q + (size_t)q_lo * qdim;
      qp[r].n_send = Qo * qdim;
      qp[r].recv = d_q + (size_t)q_bounds[r] * qdim;
      qp[r].n_recv = (q_bounds[r + 1] - q_bounds[r]) * qdim;
      if (r != rank) {
        route_send += qp[r].n_send * 4;
        route_recv += qp[r].n_recv * 4;
      }
    }
    C_HIP(hipStreamWaitEvent(rs, sl.queries_up, 0));  // this rank's own vectors are in place
    C_TRY(comm_alltoallv(s->comm, 1, qp, rs));
  }
  C_HIP(hipStreamWaitEvent(rs, s->ids_up, 0));
  C_HIP(hipMemsetAsync(d_bad, 0xFF, 8, rs));
  C_HIP(hipMemsetAsync(s->d_cnt.as<int32_t>() + (n_cnt - 1), 0, 4, rs));
  C_HIP(hipMemsetAsync(s->d_qcnt.as<int32_t>() + Q, 0, 4, rs));
  RouteBounds bounds;
  for (int r = 0; r <= world; ++r) bounds.b[r] = own_holder ? 0 : s->doc_bounds[(size_t)r];
  const int32_t* d_held = own_holder ? s->d_held.as<int32_t>() : nullptr;
  if (n_tiles > 0) {
    own_count_kernel<<<(unsigned)n_tiles, 256, 0, rs>>>(d_tile_p0, d_tile_hi, s->d_ids.as<int32_t>(), d_held, bounds, world, (int)n_tiles, s->d_cnt.as<int32_t>(), d_bad);
    C_HIP(hipGetLastError());
  }
  {
    size_t tmp = 0;
    C_HIP(rocprim::exclusive_scan(nullptr, tmp, s->d_cnt.as<int32_t>(), s->d_start.as<int32_t>(), (int32_t)0, n_cnt, rocprim::plus<int32_t>(), rs));
    C_TRY(s->d_tmp.ensure(std::max<size_t>(tmp, 16)));
    C_HIP(rocprim::exclusive_scan(s->d_tmp.p, tmp, s->d_cnt.as<int32_t>(), s->d_start.as<int32_t>(), (int32_t)0, n_cnt, rocprim::plus<int32_t>(), rs));
  }
  if (n_tiles > 0) {
    own_place_kernel<<<(unsigned)n_tiles, 256, 0, rs>>>(d_tile_p0, d_tile_hi, s->d_ids.as<int32_t>(), d_held, bounds, world, (int)n_tiles, s->d_start.as<int32_t>(), m_slot,
                                                         s->d_sendids.as<int32_t>());
    C_HIP(hipGetLastError());
  }
  own_qcount_kernel<<<(unsigned)(((int64_t)world * Qo + kMaxWorld + 255) / 256), 256, 0, rs>>>(s->d_start.as<int32_t>(), d_tile_first, world, (int)Qo, (int)n_tiles,
                                                                                            s->d_cntq.as<int32_t>(), reinterpret_cast<long long*>(d_bad));
  C_HIP(hipGetLastError());
  Piece pc[kMaxWorld];
  // exchange A: per-query counts to the holders
  for (int r = 0; r < world; ++r) {
    pc[r].send = s->d_cntq.as<int32_t>() + (size_t)r * Qo;
    pc[r].n_send = Qo;
    pc[r].recv = s->d_qcnt.as<int32_t>() + q_bounds[r];
    pc[r].n_recv = q_bounds[r + 1] - q_bounds[r];
  }
  for (int r = 0; r < world; ++r)
    if (r != rank) {
      route_send += pc[r].n_send * 4;
      route_recv += pc[r].n_recv * 4;
    }
  C_TRY(comm_alltoallv(s->comm, 1, pc, rs));
  {
    auto wide = rocprim::make_transform_iterator(s->d_qcnt.as<int32_t>(), WidenI32());
    size_t tmp = 0;
    C_HIP(rocprim::exclusive_scan(nullptr, tmp, wide, s->d_local_off.as<int64_t>(), (int64_t)0, (size_t)Q + 1, rocprim::plus<int64_t>(), rs));
    C_TRY(s->d_tmp.ensure(std::max<size_t>(tmp, 16)));
    C_HIP(rocprim::exclusive_scan(s->d_tmp.p, tmp, wide, s->d_local_off.as<int64_t>(), (int64_t)0, (size_t)Q + 1, rocprim::plus<int64_t>(), rs));
  }
  C_HIP(hipMemcpyAsync(h_small, s->d_small.p, small_words * 8, hipMemcpyDeviceToHost, rs));
  C_HIP(hipMemcpyAsync(h_local_off, s->d_local_off.p, ((size_t)Q + 1) * 8, hipMemcpyDeviceToHost, rs));
  C_HIP(hipStreamSynchronize(rs));
  const unsigned long long bad_own = (unsigned long long)h_small[0];
  sl.send_counts.assign((size_t)world, 0);
  sl.recv_counts.assign((size_t)world, 0);
  for (int r = 0; r < world; ++r) {
    sl.recv_counts[(size_t)r] = h_small[1 + r];                                         // scores coming back from holder r
    sl.send_counts[(size_t)r] = h_local_off[q_bounds[r + 1]] - h_local_off[q_bounds[r]];  // scores going to owner r
  }
  sl.local_pairs = h_local_off[Q];
  int status = FFX_OK;  // this rank's own trouble, told to the others in exchange C
  int64_t bad_local = -1;
  if (sl.local_pairs > INT32_MAX - 1) status = ffx_fail(FFX_BAD_ARG, "more than 2^31-2 candidates on one rank: split the queries");
  // exchange B: the ordinals, received straight into the plan
  int64_t* p_off = nullptr;
  int32_t* p_ids = nullptr;
  if (status == FFX_OK) status = ffx_plan_adopt_begin(s->plans[t], Q, sl.local_pairs, mode, &p_off, &p_ids);
  if (status != FFX_OK) {  // still take part: the peers are sending
    C_TRY(s->d_local_ids.ensure((size_t)std::max<int64_t>(std::min<int64_t>(sl.local_pairs, INT32_MAX), 1) * 4));
    p_ids = s->d_local_ids.as<int32_t>();
  } else {
    C_HIP(hipMemcpyAsync(p_off, s->d_local_off.p, ((size_t)Q + 1) * 8, hipMemcpyDeviceToDevice, rs));
  }
  {
    int64_t s_at = 0;
    for (int r = 0; r < world; ++r) {
      pc[r].send = s->d_sendids.as<int32_t>() + s_at;
      pc[r].n_send = sl.recv_counts[(size_t)r];
      pc[r].recv = p_ids + h_local_off[q_bounds[r]];
      pc[r].n_recv = sl.send_counts[(size_t)r];
      s_at += sl.recv_counts[(size_t)r];
      if (r != rank) {
        route_send += pc[r].n_send * 4;
        route_recv += pc[r].n_recv * 4;
      }
    }
    C_TRY(comm_alltoallv(s->comm, 1, pc, rs));
  }
  if (status == FFX_OK && own_holder && s->n_codes >= 0 && sl.local_pairs > 0) {
    codes_to_ordinals_kernel<<<(unsigned)((sl.local_pairs + 255) / 256), 256, 0, rs>>>(p_ids, sl.local_pairs, s->d_code_table.as<int32_t>(), s->n_codes);
    C_HIP(hipGetLastError());
  }
  const unsigned long long* d_plan_bad = nullptr;
  if (status == FFX_OK) status = ffx_plan_adopt_check(s->plans[t], h_local_off, rs, &d_plan_bad);  // validates on rs, asynchronously
  // exchange C, two words per peer (made on the device: nothing to wait for), so that EVERY rank leaves with the same verdict:
  //   [0] to owner r: the first of ITS ordinals that names no row here (position in its message; -1: none)
  //   [1] to everybody: this rank's status -- 0 ok, 1 one of MY candidates is held by no rank, 3 somebody's ordinal names
  //       no row here (the owner concerned finds its position in word 0), 2 anything else
  std::string own_error;
  if (status != FFX_OK) own_error = ffx_last_error();
  int32_t* d_v = s->d_verdict.as<int32_t>();
  {
    RouteBounds qbk;
    for (int r = 0; r <= world; ++r) qbk.b[r] = q_bounds[r];
    stage_verdict_kernel<<<1, kMaxWorld, 0, rs>>>(d_bad, status == FFX_OK ? d_plan_bad : nullptr, s->d_local_off.as<int64_t>(), qbk, world, status != FFX_OK ? 1 : 0, d_v);
    C_HIP(hipGetLastError());
  }
  int32_t* h_got = h_verdict + 2 * kMaxWorld;
  if (world > 1) {
    for (int r = 0; r < world; ++r) {
      pc[r].send = d_v + 2 * r;
      pc[r].n_send = 2;
      pc[r].recv = d_v + 2 * kMaxWorld + 2 * r;
      pc[r].n_recv = 2;
    }
    C_TRY(comm_alltoallv(s->comm, 1, pc, rs));
    C_HIP(hipMemcpyAsync(h_got, d_v + 2 * kMaxWorld, (size_t)world * 8, hipMemcpyDeviceToHost, rs));
  } else {
    C_HIP(hipMemcpyAsync(h_got, d_v, 8, hipMemcpyDeviceToHost, rs));
  }
  C_HIP(hipStreamSynchronize(rs));  // the second and last wait of the call
  if (s->comm->transport == FFX_COMM_REHEARSAL)  // the words moved; the peers that would have written them do not exist
    for (int r = 0; r < world; ++r)
      if (r != rank) {
        h_got[2 * r] = -1;
        h_got[2 * r + 1] = 0;
      }
  if (status == FFX_OK) status = ffx_plan_adopt_finish(s->plans[t], rs, &bad_local);
  if (status != FFX_OK && own_error.empty()) own_error = ffx_last_error();
  const bool holder_complaint = status == FFX_NOT_FOUND && bad_local >= 0;
  const int my_status = bad_own != ~0ull ? 1 : status == FFX_OK ? 0 : holder_complaint ? 3 : 2;
  s->stage_h2d_bytes = h2d;
  s->stage_route_send_bytes = route_send;
  s->stage_route_recv_bytes = route_recv;
  if (bad_own != ~0ull) {
    if (bad_pair) *bad_pair = (int64_t)bad_own;
    return ffx_fail(FFX_NOT_FOUND, "pair %llu names a document no rank of the sharded index holds", bad_own);
  }
  for (int r = 0; r < world; ++r)
    if (h_got[2 * r] >= 0) {  // holder r could not place one of THIS rank's candidates: which one?
      int64_t at = h_got[2 * r];
      for (int i = 0; i < r; ++i) at += sl.recv_counts[(size_t)i];
      std::vector<int32_t> slots((size_t)Po);
      C_TRY(ffx_d2h_now(s->device, slots.data(), m_slot, (size_t)Po * 4));
      int64_t found = -1;
      for (int64_t p = 0; p < Po && found < 0; ++p)
        if (slots[(size_t)p] == at) found = p;
      if (bad_pair) *bad_pair = found;
      return ffx_fail(FFX_NOT_FOUND, "pair %lld names an ordinal with no rows in the index (rank %d holds it)", (long long)found, r);
    }
  if (my_status == 2) return ffx_fail(status, "%s", own_error.c_str());
  for (int r = 0; r < world; ++r) {
    const int st = r == rank ? my_status : h_got[2 * r + 1];
    if (st == 1 || st == 3) return ffx_fail(FFX_NOT_FOUND, "a candidate owned by another rank names an ordinal without rows (reported by rank %d)", r);
    if (st != 0) return ffx_fail(FFX_RCCL_ERROR, "rank %d could not take the batch in", r);
  }
  C_TRY(ffx_merge_adopt_finish(s->merges[t], longest, order_base));
  C_TRY(slot_recv_room(s, t, Po));
  void* ffp = nullptr;
  C_TRY(ffx_plan_device_ptr(s->plans[t], FFX_BUF_FF, &ffp, nullptr));
  sl.send_dev = static_cast<float*>(ffp);
  sl.own[0] = q_lo;
  sl.own[1] = q_lo + Qo;
  sl.own[2] = 0;
  sl.own[3] = Po;
  sl.have_sparse = own_sparse != nullptr;
  sl.merge = t;
  C_HIP(hipEventRecord(s->r_end, rs));
  C_HIP(hipStreamWaitEvent(rs, sl.queries_up, 0));
  C_HIP(hipStreamWaitEvent(rs, sl.sparse_up, 0));
  C_HIP(hipEventRecord(sl.routed, rs));
  sl.routed_valid = true;
  sl.scored_valid = sl.sent_valid = sl.merged_valid = false;
  s->out_valid[t] = false;
  for (int k = 0; k < 4; ++k) s->own[k] = sl.own[k];
  s->local_pairs = sl.local_pairs;
  s->have_sparse = sl.have_sparse;
  s->fresh = true;
  s->cur = t;
  ++s->staged;
  s->have_pairs = true;
  // the caller's own_* arrays are free again on return (ffx.h): ordinary memory has been consumed by ffx_h2d, page-locked
  // memory is read by copies on the upload stream, of which the first-stage scores are the last (normally long done:
  // the two waits for the routing stream above came after the ordinals, which precede them on the same stream)
  C_HIP(hipEventSynchronize(sl.sparse_up));
  s->stage_host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
  return FFX_OK;
}

}  // namespace

extern "C" {

int ffx_sharded_create(ffx_index* shard, ffx_comm* comm, const int64_t* doc_bounds, ffx_sharded** out) {
  if (!shard || !comm || !out) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  *out = nullptr;
  if (comm->device < 0) return ffx_fail(FFX_BAD_ARG, "the communicator has no device");
  ffx_sharded* s = new (std::nothrow) ffx_sharded();
  if (!s) return ffx_fail(FFX_OOM, "host allocation failed");
  s->shard = shard;
  s->comm = comm;
  s->device = comm->device;
  if (doc_bounds) {
    s->doc_bounds.assign(doc_bounds, doc_bounds + comm->world + 1);
    for (int r = 0; r < comm->world; ++r)
      if (doc_bounds[r + 1] < doc_bounds[r]) {
        delete s;
        return ffx_fail(FFX_BAD_ARG, "doc_bounds not monotone at rank %d", r);
      }
  }
  int rc = FFX_OK;
  hipError_t e = hipSetDevice(s->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->compute, hipStreamNonBlocking);
  for (int i = 0; i < kSlots && e == hipSuccess; ++i) {
    for (hipEvent_t* ev : {&s->slot[i].routed, &s->slot[i].scored, &s->slot[i].sent, &s->slot[i].merged, &s->slot[i].queries_up, &s->slot[i].sparse_up, &s->out_done[i]})
      if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
  }
  for (hipEvent_t* ev : {&s->x_begin, &s->x_end, &s->m_begin, &s->m_end, &s->r_begin, &s->r_end})
    if (e == hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableSystemFence);  // timing only: no cache write-back when recorded
  if (e != hipSuccess) rc = ffx_fail(FFX_HIP_ERROR, "stream/event creation failed: %s", hipGetErrorString(e));
  // the plans issue their work on the compute stream and own none (a process has few hardware queues: every stream
  // that is created and never used still takes part in their assignment)
  for (int i = 0; i < kSlots && rc == FFX_OK; ++i) {
    rc = ffx_plan_create_on(shard, s->compute, &s->plans[i]);
    if (rc == FFX_OK) rc = ffx_merge_create_on(s->device, &s->merges[i]);
  }
  if (rc != FFX_OK) {
    ffx_sharded_destroy(s);
    return rc;
  }
  *out = s;
  return FFX_OK;
}

int ffx_sharded_destroy(ffx_sharded* s) {
  if (!s) return FFX_OK;
  (void)hipSetDevice(s->device);
  if (s->compute) (void)hipStreamSynchronize(s->compute);
  if (s->comm && s->comm->stream) (void)hipStreamSynchronize(s->comm->stream);
  if (s->comm && s->comm->route_stream) (void)hipStreamSynchronize(s->comm->route_stream);
  if (s->out) (void)hipStreamSynchronize(s->out);
  if (s->up) (void)hipStreamSynchronize(s->up);
  for (auto* p : s->plans)
    if (p) ffx_plan_destroy(p);
  for (auto* m : s->merges)
    if (m) ffx_merge_destroy(m);
  for (auto* a : {&s->d_ids, &s->d_off, &s->d_who, &s->d_cnt, &s->d_local_off, &s->d_local_ids, &s->d_pos, &s->d_keys, &s->d_arrival, &s->d_slot,
                  &s->d_small, &s->d_tmp, &s->d_held, &s->d_start, &s->d_sendids, &s->d_qcnt, &s->d_verdict, &s->d_cntq, &s->d_tiles, &s->d_code_table})
    a->release();
  if (s->host_small) (void)hipHostFree(s->host_small);
  for (int i = 0; i < kSlots; ++i) {
    for (hipEvent_t ev : {s->slot[i].routed, s->slot[i].scored, s->slot[i].sent, s->slot[i].merged, s->slot[i].queries_up, s->slot[i].sparse_up, s->out_done[i]})
      if (ev) (void)hipEventDestroy(ev);
    if (s->slot[i].recv) (void)hipFree(s->slot[i].recv);
  }
  for (hipEvent_t ev : {s->x_begin, s->x_end, s->m_begin, s->m_end, s->r_begin, s->r_end})
    if (ev) (void)hipEventDestroy(ev);
  if (s->ids_up) (void)hipEventDestroy(s->ids_up);
  if (s->out) (void)hipStreamDestroy(s->out);
  if (s->up) (void)hipStreamDestroy(s->up);
  if (s->compute) (void)hipStreamDestroy(s->compute);
  delete s;
  return FFX_OK;
}

int ffx_sharded_set_queries(ffx_sharded* s, const float* host_queries, int64_t n_queries) {
  if (!s) return ffx_fail(FFX_BAD_ARG, "NULL handle");
  C_TRY(ffx_sharded_sync(s));
  for (auto* p : s->plans) C_TRY(ffx_plan_set_queries(p, host_queries, n_queries));  // replicated: Q*d*4 bytes
  return FFX_OK;
}

int ffx_sharded_set_pairs(ffx_sharded* s, const int64_t* pair_off, int64_t n_queries, const int32_t* pair_ids, const float* sparse,
                          int mode, int64_t* bad_pair) {
  if (s && s->doc_bounds.empty()) return ffx_fail(FFX_BAD_ARG, "created without doc_bounds: use ffx_sharded_set_pairs_held");
  return sharded_set(s, pair_off, n_queries, pair_ids, nullptr, sparse, mode, bad_pair);
}

int ffx_sharded_set_pairs_held(ffx_sharded* s, const int64_t* pair_off, int64_t n_queries, const int32_t* holder,
                               const int32_t* local_ids, const float* sparse, int mode, int64_t* bad_pair) {
  if (!holder) return ffx_fail(FFX_BAD_ARG, "holder is NULL");
  return sharded_set(s, pair_off, n_queries, local_ids, holder, sparse, mode, bad_pair);
}

int ffx_sharded_set_code_table(ffx_sharded* s, const int32_t* ordinal_of_code, int64_t n_codes) {
  if (!s || n_codes < 0) return ffx_fail(FFX_BAD_ARG, "bad argument");
  C_HIP(hipSetDevice(s->device));
  C_TRY(ffx_sharded_sync(s));
  if (s->comm->route_stream) C_HIP(hipStreamSynchronize(s->comm->route_stream));
  if (!ordinal_of_code) {
    s->n_codes = -1;
    return FFX_OK;
  }
  C_TRY(s->d_code_table.ensure((size_t)std::max<int64_t>(n_codes, 1) * 4));
  if (n_codes > 0) C_TRY(ffx_h2d_now(s->device, s->d_code_table.p, ordinal_of_code, (size_t)n_codes * 4));
  s->n_codes = n_codes;
  return FFX_OK;
}

int ffx_sharded_stage_queries(ffx_sharded* s, const float* all_queries, int64_t n_queries) {
  if (!s || n_queries < 0 || (n_queries > 0 && !all_queries)) return ffx_fail(FFX_BAD_ARG, "bad argument");
  C_HIP(hipSetDevice(s->device));
  if (!s->fresh) {  // coming from the set-once form: nothing of it may still be in flight
    C_TRY(ffx_sharded_sync(s));
    for (auto& sl : s->slot) sl.routed_valid = sl.scored_valid = sl.sent_valid = sl.merged_valid = false;
    for (bool& v : s->out_valid) v = false;
    s->last_merged = -1;
    s->fresh = true;
    s->have_pairs = false;
  }
  const int t = (int)(s->staged % kSlots);
  ffx_sharded::Slot& sl = s->slot[t];
  C_TRY(sharded_io_streams(s));
  if (sl.scored_valid) C_HIP(hipStreamWaitEvent(s->up, sl.scored, 0));  // the batch that read this slot's vectors last has been scored
  float* d_q = nullptr;
  C_TRY(ffx_plan_queries_room(s->plans[t], n_queries, &d_q));
  int32_t qdim = 0;
  C_TRY(ffx_index_query_dim(s->shard, &qdim));
  if (n_queries > 0) C_TRY(ffx_h2d(s->device, d_q, all_queries, (size_t)n_queries * qdim * 4, s->up));
  s->staged_query_bytes = n_queries * (int64_t)qdim * 4;
  return FFX_OK;
}

int ffx_sharded_stage_owned(ffx_sharded* s, const int64_t* q_bounds, const float* own_queries, const int64_t* own_pair_off,
                            const int32_t* own_pair_ids, const int32_t* own_holder, const float* own_sparse, int mode, int64_t* bad_pair) {
  return sharded_stage_owned(s, q_bounds, own_queries, own_pair_off, own_pair_ids, own_holder, own_sparse, mode, 0, bad_pair);
}

// The owner-side half of batch `s->pending` (scatter to first-stage order, interpolation, per-query order), once that
// batch's scores have arrived.  Two placements (by size; FFX_DEBUG_VARIANT=merge_on_compute / merge_on_results force one):
//   compute stream  behind whatever has been queued there: never beside a gather_dot, but every step then pays for it
//                   (merge 0.03 + order 0.07 ms + two launch gaps between one gather_dot and the next at an eighth of the batch)
//   result stream   (high priority, where the results leave anyway): its waves take the slots gather_dot's retiring waves
//                   free, so it runs INSIDE the next batch's gather_dot instead of between two of them
static int sharded_merge_pending(ffx_sharded* s) {
  if (s->pending < 0) return FFX_OK;
  const int i = s->pending;
  s->pending = -1;
  ffx_sharded::Slot& sl = s->slot[i];
  hipStream_t ms = s->compute;
  // a small merge hides inside the next gather_dot; a large one (tens of millions of candidates: one or two ranks) takes
  // slots from it for its whole length and is better paid for once, in line (measured: HISTORY.md section 7)
  const bool small = sl.own[3] <= (int64_t)(6 << 20);  // (4.4 M owned candidates: hidden, 2.26 -> 2.12 ms; 8.7 M: the fresh step loses 5 %; 34.9 M: 7 %)
  if (ffx_experiment("merge_on_results", nullptr) || (small && !ffx_experiment("merge_on_compute", nullptr))) {
    C_TRY(sharded_io_streams(s));
    ms = s->out;
  }
  C_HIP(hipStreamWaitEvent(ms, sl.sent, 0));
  if (ms != s->out && s->out_valid[sl.merge]) C_HIP(hipStreamWaitEvent(ms, s->out_done[sl.merge], 0));  // the results this merge held before have left
  C_HIP(hipEventRecord(s->m_begin, ms));
  C_TRY(ffx_merge_run(s->merges[sl.merge], static_cast<const float*>(sl.recv), s->pending_alpha, s->pending_flags & (FFX_FLAG_SORT | FFX_FLAG_SORT_FF), ms));
  C_HIP(hipEventRecord(s->m_end, ms));
  C_HIP(hipEventRecord(sl.merged, ms));
  sl.merged_valid = true;
  s->last_merged = i;
  return FFX_OK;
}

int ffx_sharded_run(ffx_sharded* s, double alpha, uint32_t flags) {
  if (!s) return ffx_fail(FFX_BAD_ARG, "NULL handle");
  if (!s->have_pairs) return ffx_fail(FFX_BAD_ARG, "no batch has been set on this handle (ffx_sharded_set_pairs / ffx_sharded_stage_owned)");
  C_HIP(hipSetDevice(s->device));
  const int i = s->fresh ? s->cur : (int)(s->runs & 1);
  ++s->runs;
  s->last = i;
  ffx_sharded::Slot& sl = s->slot[i];
  hipStream_t cs = s->comm->stream;
  // Everything that needs the CUs runs in order on the compute stream; only the exchange -- link-bound, a handful of
  // workgroups -- runs beside it on the communicator's stream (and the routing of the NEXT batch on the routing lane):
  //   compute:  expand, gather_dot (batch i)   | merge + order (batch i-1) | expand, gather_dot (batch i+1) | ...
  //   comm:                                      exchange (batch i)                                          ...
  //   route:    ids of batch i+1 -> holders                                  ids of batch i+2 -> holders      ...
  // so the exchange of batch i is hidden behind the merge of batch i-1 and the scoring of batch i+1, and the merge
  // of batch i-1 never shares the CUs with a gather_dot.  (Round 2 ran merge + order on the communicator's stream
  // NEXT TO the following gather_dot: beside a kernel that fills every CU the LDS-heavy ordering got no slots and
  // stretched from 0.07 ms to the gather's whole 2.0 ms at an eighth of the batch, and the next exchange queued
  // behind it: profiles/r03/rehearsal_trace_n8_before.csv.)
  if (sl.routed_valid) C_HIP(hipStreamWaitEvent(s->compute, sl.routed, 0));  // the lists, the queries and the owner-side state are in place
  if (sl.sent_valid) C_HIP(hipStreamWaitEvent(s->compute, sl.sent, 0));      // the exchange that read this plan's scores last has finished
  if (s->pending == i) C_TRY(sharded_merge_pending(s));                      // (the same slot run twice in a row: its merge first)
  C_TRY(ffx_plan_run(s->plans[i], 0.0, flags & FFX_FLAG_EXACT_SUM));
  C_HIP(hipEventRecord(sl.scored, s->compute));
  sl.scored_valid = true;
  C_HIP(hipStreamWaitEvent(cs, sl.scored, 0));
  if (sl.merged_valid) C_HIP(hipStreamWaitEvent(cs, sl.merged, 0));  // the merge that read this slot's receive buffer last has finished
  C_HIP(hipEventRecord(s->x_begin, cs));
  C_TRY(ffx_comm_exchange(s->comm, sl.send_dev, sl.send_counts.data(), static_cast<float*>(sl.recv), sl.recv_counts.data(), cs));
  C_HIP(hipEventRecord(s->x_end, cs));
  C_HIP(hipEventRecord(sl.sent, cs));
  sl.sent_valid = true;
  C_TRY(sharded_merge_pending(s));  // batch i-1, behind batch i's gather_dot
  s->pending = i;
  s->pending_alpha = alpha;
  s->pending_flags = flags;
  return FFX_OK;
}

int ffx_sharded_sync(ffx_sharded* s) {
  if (!s) return ffx_fail(FFX_BAD_ARG, "NULL handle");
  C_HIP(hipSetDevice(s->device));
  C_TRY(sharded_merge_pending(s));  // the last batch has nothing behind it to hide its merge
  C_HIP(hipStreamSynchronize(s->comm->stream));
  C_HIP(hipStreamSynchronize(s->compute));
  if (s->out) C_HIP(hipStreamSynchronize(s->out));
  return FFX_OK;
}

int ffx_sharded_fetch(ffx_sharded* s, float* out_ff, float* out_interp, int32_t* out_order) {
  if (!s) return ffx_fail(FFX_BAD_ARG, "NULL handle");
  C_TRY(ffx_sharded_sync(s));
  if (s->last_merged < 0) return ffx_fail(FFX_BAD_ARG, "ffx_sharded_run has not been called since the batch was set");
  return ffx_merge_fetch(s->merges[s->slot[s->last_merged].merge], out_ff, out_interp, out_order);
}

int ffx_sharded_fetch_async(ffx_sharded* s, float* out_ff, float* out_interp, int32_t* out_order) {
  if (!s) return ffx_fail(FFX_BAD_ARG, "NULL handle");
  if (s->last_merged < 0) return ffx_fail(FFX_BAD_ARG, "no batch has been merged yet: ffx_sharded_run queues the merge of the batch BEFORE the one it runs");
  C_HIP(hipSetDevice(s->device));
  C_TRY(sharded_io_streams(s));
  ffx_sharded::Slot& sl = s->slot[s->last_merged];
  C_HIP(hipStreamWaitEvent(s->out, sl.merged, 0));
  C_TRY(ffx_merge_fetch_async(s->merges[sl.merge], out_ff, out_interp, out_order, s->out));
  C_HIP(hipEventRecord(s->out_done[sl.merge], s->out));
  s->out_valid[sl.merge] = true;
  return FFX_OK;
}

int ffx_sharded_owned(const ffx_sharded* s, int64_t* q_lo, int64_t* q_hi, int64_t* first_pair, int64_t* n_pairs) {
  if (!s) return ffx_fail(FFX_BAD_ARG, "NULL handle");
  if (!s->have_pairs) return ffx_fail(FFX_BAD_ARG, "no batch has been set on this handle");
  if (q_lo) *q_lo = s->own[0];
  if (q_hi) *q_hi = s->own[1];
  if (first_pair) *first_pair = s->own[2];
  if (n_pairs) *n_pairs = s->own[3];
  return FFX_OK;
}

// One batch from host buffers to host buffers, every rank handing over the queries it owns: the block is cut into
// n_slices slices of whole queries (the same number on every rank; slice j of every rank together is sub-batch j) and the
// sub-batches go through the three slots -- while sub-batch j is scored, j+1 is routed and the results of j-1 leave.
int ffx_sharded_score(ffx_sharded* s, const int64_t* q_bounds, const float* own_queries, const int64_t* own_pair_off, const int32_t* own_pair_ids,
                      const int32_t* own_holder, const float* own_sparse, int mode, double alpha, uint32_t flags, int32_t n_slices, float* out_ff,
                      float* out_interp, int32_t* out_order, int64_t* bad_pair) {
  if (bad_pair) *bad_pair = -1;
  if (!s || !q_bounds || !own_pair_off) return ffx_fail(FFX_BAD_ARG, "bad argument");
  if (out_interp && !own_sparse) return ffx_fail(FFX_BAD_ARG, "no sparse scores were given: nothing to interpolate");
  if (out_order && !(flags & FFX_FLAG_SORT)) return ffx_fail(FFX_BAD_ARG, "out_order needs FFX_FLAG_SORT");
  int rank = 0, world = 1;
  ffx_comm_rank(s->comm, &rank, &world);
  const int64_t Qo = q_bounds[rank + 1] - q_bounds[rank];
  if (Qo < 0) return ffx_fail(FFX_BAD_ARG, "q_bounds not monotone");
  if (n_slices <= 0) {  // by size: ~4 M pairs per full slice on the rank that has most (every rank must cut alike)
    double most = (double)own_pair_off[Qo];
    C_TRY(ffx_comm_allreduce_f64(s->comm, &most, 1, FFX_OP_MAX));
    const int full = (int)std::min<double>(64.0, std::max(1.0, std::ceil(most / (double)(4 << 20))));
    n_slices = full >= 3 ? full + 4 : full;  // room for the ramp below
  }
  // The first slices are small (1/8, 1/4, 1/2 of a full one: the device idles while the first one is taken in) and so are
  // the last (the caller idles while the last one's results come back), as in ffx_score.  Weights by slice NUMBER only:
  // every rank cuts its own block at the same fractions.
  std::vector<int64_t> weight_before((size_t)n_slices + 1, 0);
  for (int j = 0; j < n_slices; ++j) {
    int w = 8;
    if (n_slices >= 7) {
      const int from_end = n_slices - 1 - j;
      w = std::min(8, std::min(1 << std::min(j, 3), 1 << std::min(from_end, 3)));
    }
    weight_before[(size_t)j + 1] = weight_before[(size_t)j] + w;
  }
  int32_t qdim = 0;
  C_TRY(ffx_index_query_dim(s->shard, &qdim));
  std::vector<int64_t> qb((size_t)world + 1), off;
  // first query of rank r's slice j, in its block
  auto cut = [&](int r, int j) { return (int64_t)(((__int128)(q_bounds[r + 1] - q_bounds[r]) * weight_before[(size_t)j]) / weight_before[(size_t)n_slices]); };
  int rc = FFX_OK;
  int64_t first_pair[2] = {0, 0};  // of the slice whose results leave next
  for (int j = 0; j <= n_slices && rc == FFX_OK; ++j) {
    if (j < n_slices) {
      qb[0] = 0;
      for (int r = 0; r < world; ++r) qb[(size_t)r + 1] = qb[(size_t)r] + cut(r, j + 1) - cut(r, j);
      const int64_t q0 = cut(rank, j), q1 = cut(rank, j + 1), p0 = own_pair_off[q0];
      off.resize((size_t)(q1 - q0) + 1);
      for (int64_t q = q0; q <= q1; ++q) off[(size_t)(q - q0)] = own_pair_off[q] - p0;
      int64_t bad = -1;
      rc = sharded_stage_owned(s, qb.data(), own_queries ? own_queries + (size_t)q0 * qdim : nullptr, off.data(), own_pair_ids ? own_pair_ids + p0 : nullptr,
                               own_holder ? own_holder + p0 : nullptr, own_sparse ? own_sparse + p0 : nullptr, mode, (int32_t)p0, &bad);
      if (rc != FFX_OK) {
        if (bad_pair && bad >= 0) *bad_pair = p0 + bad;
        break;
      }
      rc = ffx_sharded_run(s, alpha, flags);  // queues the merge of slice j-1 behind slice j's gather_dot
      first_pair[j & 1] = p0;
    } else {
      rc = sharded_merge_pending(s);  // the last slice has nothing behind it
    }
    if (rc == FFX_OK && j >= 1) {
      const int64_t p0 = first_pair[(j - 1) & 1];
      rc = ffx_sharded_fetch_async(s, out_ff ? out_ff + p0 : nullptr, out_interp ? out_interp + p0 : nullptr, out_order ? out_order + p0 : nullptr);
    }
  }
  if (rc != FFX_OK) {  // nothing may still be writing when the caller gets its buffers back
    const std::string keep = ffx_last_error();
    s->pending = -1;
    (void)hipStreamSynchronize(s->comm->stream);
    (void)hipStreamSynchronize(s->compute);
    if (s->out) (void)hipStreamSynchronize(s->out);
    ffx_fail(rc, "%s", keep.c_str());
    return rc;
  }
  return ffx_sharded_sync(s);
}

int ffx_sharded_stats(ffx_sharded* s, ffx_shard_stats* out) {
  if (!s || !out) return ffx_fail(FFX_BAD_ARG, "NULL argument");
  if (s->runs == 0) return ffx_fail(FFX_BAD_ARG, "ffx_sharded_run has not been called");
  C_TRY(ffx_sharded_sync(s));
  memset(out, 0, sizeof *out);
  const ffx_sharded::Slot& sl = s->slot[s->last];
  C_TRY(ffx_plan_stats(s->plans[s->last], &out->local));
  ffx_comm_rank(s->comm, &out->rank, &out->world);
  out->own_queries = sl.own[1] - sl.own[0];
  out->own_pairs = sl.own[3];
  for (int r = 0; r < out->world; ++r) {
    if (r == out->rank) continue;
    out->send_bytes += sl.send_counts[(size_t)r] * 4;
    out->recv_bytes += sl.recv_counts[(size_t)r] * 4;
  }
  out->transport = s->comm->transport;
  C_HIP(hipEventElapsedTime(&out->ms_exchange, s->x_begin, s->x_end));
  C_HIP(hipEventElapsedTime(&out->ms_merge, s->m_begin, s->m_end));
  if (s->fresh) {
    out->stage_h2d_bytes = s->stage_h2d_bytes;
    out->stage_route_send_bytes = s->stage_route_send_bytes;
    out->stage_route_recv_bytes = s->stage_route_recv_bytes;
    out->ms_stage_host = (float)s->stage_host_ms;
    if (s->comm->route_stream) {
      C_HIP(hipStreamSynchronize(s->comm->route_stream));
      C_HIP(hipEventElapsedTime(&out->ms_stage_device, s->r_begin, s->r_end));
    }
  }
  return FFX_OK;
}

int ffx_sharded_device_ptr(ffx_sharded* s, int which, void** ptr, int64_t* n_elems) {
  if (!s) return ffx_fail(FFX_BAD_ARG, "NULL handle");
  // what ffx_sharded_fetch would copy out: the batch merged last.  (The merge of the batch run last is still pending
  // until ffx_sharded_sync: a caller that wants ITS results syncs first.)
  const int m = s->last_merged >= 0 ? s->slot[s->last_merged].merge : 0;
  return ffx_merge_device_ptr(s->merges[m], which, ptr, n_elems);
}

}  // extern "C"
