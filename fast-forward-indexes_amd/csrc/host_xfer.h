// host_xfer.h -- how bytes move between the CALLER'S host memory and the device (shared by ffx.hip and comm.hip).
//
// A transfer of more than 1 MiB from or into ordinary (pageable) memory makes the HIP runtime page-lock the caller's pages
// on the fly and release them afterwards.  This library never lets it: such a transfer is cut into pieces of 1 - 8 MiB that travel
// through page-locked blocks the library owns, filled (or emptied) by a few host threads while the pieces before them are on
// the wire; the pieces are queued on the caller's stream (no stream is created here).  Memory the caller page-locked itself (ffx_host_alloc,
// ffx_host_register, hipHostMalloc) and transfers of at most 1 MiB (the runtime stages those through its own blocks, it
// never pins for them) go to the runtime as they are.  Nothing here changes a setting of the runtime or of the process.
//
//   ffx_h2d(device, dst, src, bytes, stream)   the source is consumed on return; the bytes are on the device once
//                                              `stream` reaches this point (stream order, like hipMemcpyAsync)
//   ffx_d2h(device, dst, src, bytes, stream)   ordered behind what `stream` holds now; on return the bytes are either in
//                                              dst (staged form) or queued on `stream` (direct form): synchronise
//                                              `stream` before reading, as after hipMemcpyAsync
#ifndef FFX_HOST_XFER_H_
#define FFX_HOST_XFER_H_

#include <hip/hip_runtime.h>

#include <cstddef>

// true if the runtime knows `p` as page-locked host memory (or as device memory): it may be handed to hipMemcpyAsync as is
bool ffx_runtime_owns(const void* p);
int ffx_h2d(int device, void* dst, const void* src, size_t bytes, hipStream_t stream);
int ffx_d2h(int device, void* dst, const void* src, size_t bytes, hipStream_t stream);
// several arrays of one batch at once: their staged pieces share one round of threads (same guarantees as ffx_h2d for each)
int ffx_h2d_many(int device, int n_parts, void* const* dsts, const void* const* srcs, const size_t* sizes, hipStream_t stream);
// the blocking forms (what hipMemcpy on the null stream was): done on return
int ffx_h2d_now(int device, void* dst, const void* src, size_t bytes);
int ffx_d2h_now(int device, void* dst, const void* src, size_t bytes);
// page-locked blocks and streams of every device's lanes are released (ffx_host_drain / process exit paths; lanes come back on demand)
void ffx_xfer_release();
// counters for tests and tools: transfers and bytes that went through the library's blocks since the process began
void ffx_xfer_counters(unsigned long long* staged_transfers, unsigned long long* staged_bytes);

#endif  // FFX_HOST_XFER_H_
