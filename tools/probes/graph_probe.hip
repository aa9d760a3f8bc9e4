// Does a hipGraph shorten a launch-bound sequence on this stack?  N small dependent kernels on one stream + one wait,
// issued kernel by kernel vs as one instantiated graph.  (hipcc --offload-arch=gfx950 -O2 graph_probe.hip -o graph_probe)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void step(float* p, int n, float a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = p[i] * a + 1.0f;
}
int main() {
  const int n = 1 << 16, K = 6, iters = 2000;
  float* d = nullptr;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemset(d, 0, n * sizeof(float)));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  auto enqueue = [&]() {
    for (int k = 0; k < K; ++k) step<<<n / 256, 256, 0, st>>>(d, n, 0.5f);
  };
  for (int w = 0; w < 50; ++w) enqueue();
  CK(hipStreamSynchronize(st));
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) { enqueue(); CK(hipStreamSynchronize(st)); }
  double us_plain = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  enqueue();
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int w = 0; w < 50; ++w) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) { CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st)); }
  double us_graph = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
  // back to back without waiting (throughput of issue)
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) enqueue();
  CK(hipStreamSynchronize(st));
  double us_plain_b2b = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
  t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  double us_graph_b2b = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
  printf("{\"kernels\": %d, \"plain_us_per_iter_waited\": %.2f, \"graph_us_per_iter_waited\": %.2f, \"plain_us_per_iter_back_to_back\": %.2f, \"graph_us_per_iter_back_to_back\": %.2f}\n",
         K, us_plain, us_graph, us_plain_b2b, us_graph_b2b);
  return 0;
}
