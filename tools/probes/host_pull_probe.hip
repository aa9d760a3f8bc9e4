// Can the GPU pull index rows straight out of a memory-mapped index file (page cache / tmpfs), without host threads
// copying them into a staging block first?  Registers (a) a file-backed read-only mapping and (b) page-locked anonymous
// memory, and times a kernel that copies N random 3072-byte rows from each into HBM, beside a plain pinned hipMemcpy.
//   hipcc --offload-arch=gfx950 -O2 host_pull_probe.hip -o host_pull_probe ; ./host_pull_probe /dev/shm/probe.bin
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// one wave per row: 64 lanes x 16 B x 3 = 3072 B
__global__ void __launch_bounds__(256) pull_rows(const uint4* __restrict__ src, const int64_t* __restrict__ rows, int n, uint4* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6;
  for (int r = wave; r < n; r += n_waves) {
    const uint4* s = src + rows[r] * 192;
    uint4* d = dst + (size_t)r * 192;
    const uint4 a = s[lane], b = s[64 + lane], c = s[128 + lane];
    d[lane] = a;
    d[64 + lane] = b;
    d[128 + lane] = c;
  }
}

static double time_pull(const void* dev_src, const int64_t* d_rows, int n, uint4* d_dst, int grid) {
  for (int w = 0; w < 2; ++w) pull_rows<<<grid, 256>>>((const uint4*)dev_src, d_rows, n, d_dst);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 5; ++i) pull_rows<<<grid, 256>>>((const uint4*)dev_src, d_rows, n, d_dst);
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 5;
}

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "/dev/shm/host_pull_probe.bin";
  const size_t row_bytes = 3072, n_file_rows = 1000000, bytes = row_bytes * n_file_rows;  // 3 GB
  const int n = 170000;                                                                 // one 512 MB window
  {
    int fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd < 0 || ftruncate(fd, bytes) != 0) { perror("file"); return 1; }
    void* w = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (w == MAP_FAILED) { perror("mmap rw"); return 1; }
    for (size_t r = 0; r < n_file_rows; ++r) memset((char*)w + r * row_bytes, (int)(r & 0xff), row_bytes);
    munmap(w, bytes);
    close(fd);
  }
  std::mt19937_64 rng(1);
  std::vector<int64_t> rows(n);
  for (auto& r : rows) r = (int64_t)(rng() % n_file_rows);
  std::sort(rows.begin(), rows.end());
  int64_t* d_rows;
  uint4* d_dst;
  CK(hipMalloc(&d_rows, n * sizeof(int64_t)));
  CK(hipMalloc(&d_dst, (size_t)n * row_bytes));
  CK(hipMemcpy(d_rows, rows.data(), n * sizeof(int64_t), hipMemcpyHostToDevice));
  const double gb = (double)n * row_bytes / 1e9;
  // control: plain pinned copy
  {
    void* pinned;
    CK(hipHostMalloc(&pinned, (size_t)n * row_bytes, hipHostMallocDefault));
    memset(pinned, 1, (size_t)n * row_bytes);
    CK(hipMemcpy(d_dst, pinned, (size_t)n * row_bytes, hipMemcpyHostToDevice));
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 5; ++i) CK(hipMemcpy(d_dst, pinned, (size_t)n * row_bytes, hipMemcpyHostToDevice));
    printf("{\"pinned_memcpy_GBs\": %.2f}\n", gb * 5 / std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    CK(hipHostFree(pinned));
  }
  // (b) anonymous page-locked memory holding the whole "file", pulled by the kernel
  {
    void* anon;
    CK(hipHostMalloc(&anon, bytes, hipHostMallocDefault));
    for (size_t r = 0; r < n_file_rows; ++r) memset((char*)anon + r * row_bytes, (int)(r & 0xff), row_bytes);
    for (int grid : {4, 8, 16, 32, 64, 256, 1024, 4096}) {
      const double s = time_pull(anon, d_rows, n, d_dst, grid);
      printf("{\"source\": \"hipHostMalloc\", \"grid\": %d, \"pull_GBs\": %.2f}\n", grid, s > 0 ? gb / s : -1.0);
    }
    // what a pull of each size does to an HBM-bound kernel running beside it on another stream (the same row copy, HBM
    // to HBM, 20 launches back to back), and what that kernel does to the pull
    {
      uint4 *v_src, *v_dst;
      CK(hipMalloc(&v_src, (size_t)n * row_bytes));
      CK(hipMalloc(&v_dst, (size_t)n * row_bytes));
      CK(hipMemset(v_src, 1, (size_t)n * row_bytes));
      std::vector<int64_t> ident(n);
      for (int i = 0; i < n; ++i) ident[i] = (int64_t)(rng() % n);
      int64_t* d_ident;
      CK(hipMalloc(&d_ident, n * sizeof(int64_t)));
      CK(hipMemcpy(d_ident, ident.data(), n * sizeof(int64_t), hipMemcpyHostToDevice));
      hipStream_t s_pull, s_victim;
      CK(hipStreamCreateWithFlags(&s_pull, hipStreamNonBlocking));
      CK(hipStreamCreateWithFlags(&s_victim, hipStreamNonBlocking));
      hipEvent_t e0, e1, p0, p1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&p0)); CK(hipEventCreate(&p1));
      auto victim = [&](float* ms) {
        CK(hipEventRecord(e0, s_victim));
        for (int i = 0; i < 20; ++i) pull_rows<<<2048, 256, 0, s_victim>>>(v_src, d_ident, n, v_dst);
        CK(hipEventRecord(e1, s_victim));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(ms, e0, e1));
        return 0;
      };
      float alone = 0;
      if (victim(&alone)) return 1;
      if (victim(&alone)) return 1;
      printf("{\"victim_alone_ms_per_launch\": %.4f}\n", alone / 20);
      for (int grid : {4, 8, 16, 32, 64, 256, 1024}) {
        float beside = 0, pull_ms = 0;
        CK(hipEventRecord(p0, s_pull));
        for (int i = 0; i < 3; ++i) pull_rows<<<grid, 256, 0, s_pull>>>((const uint4*)anon, d_rows, n, d_dst);
        CK(hipEventRecord(p1, s_pull));
        if (victim(&beside)) return 1;
        CK(hipEventSynchronize(p1));
        CK(hipEventElapsedTime(&pull_ms, p0, p1));
        printf("{\"pull_grid\": %d, \"pull_GBs_beside_victim\": %.2f, \"victim_ms_per_launch_beside_pull\": %.4f}\n", grid, gb * 3 / (pull_ms * 1e-3), beside / 20);
      }
      CK(hipDeviceSynchronize());
    }
    std::vector<unsigned char> back(row_bytes);
    CK(hipMemcpy(back.data(), (char*)d_dst + (size_t)(n - 1) * row_bytes, row_bytes, hipMemcpyDeviceToHost));
    printf("{\"check_anon\": %s}\n", back[0] == (unsigned char)(rows[n - 1] & 0xff) && back[3071] == back[0] ? "true" : "false");
    CK(hipHostFree(anon));
  }
  // (a) the file itself, mapped read-only and registered
  {
    int fd = open(path, O_RDONLY);
    void* m = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) { perror("mmap ro"); return 1; }
    struct { const char* name; unsigned flags; } tries[] = {{"hipHostRegisterDefault", hipHostRegisterDefault},
                                                            {"hipHostRegisterReadOnly", hipHostRegisterReadOnly},
                                                            {"hipHostRegisterMapped|ReadOnly", hipHostRegisterMapped | hipHostRegisterReadOnly}};
    bool ok = false;
    for (auto& t : tries) {
      auto t0 = std::chrono::steady_clock::now();
      hipError_t e = hipHostRegister(m, bytes, t.flags);
      const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      printf("{\"register\": \"%s\", \"result\": \"%s\", \"seconds\": %.3f}\n", t.name, hipGetErrorString(e), s);
      if (e == hipSuccess) { ok = true; break; }
      (void)hipGetLastError();
    }
    if (ok) {
      void* dev = nullptr;
      CK(hipHostGetDevicePointer(&dev, m, 0));
      CK(hipMemset(d_dst, 0, (size_t)n * row_bytes));
      for (int grid : {256, 1024, 4096}) {
        const double s = time_pull(dev, d_rows, n, d_dst, grid);
        printf("{\"source\": \"registered read-only file mapping\", \"grid\": %d, \"pull_GBs\": %.2f}\n", grid, s > 0 ? gb / s : -1.0);
      }
      std::vector<unsigned char> back(row_bytes);
      CK(hipMemcpy(back.data(), (char*)d_dst + (size_t)(n - 1) * row_bytes, row_bytes, hipMemcpyDeviceToHost));
      printf("{\"check_file\": %s}\n", back[0] == (unsigned char)(rows[n - 1] & 0xff) && back[3071] == back[0] ? "true" : "false");
      auto t0 = std::chrono::steady_clock::now();
      CK(hipHostUnregister(m));
      printf("{\"unregister_seconds\": %.3f}\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    munmap(m, bytes);
    close(fd);
  }
  unlink(path);
  return 0;
}
