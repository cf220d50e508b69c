// GPU probe (not product code): throughput of chains of dependent small kernels issued from K host threads, each on
// its own stream: K x hipLaunchKernelGGL chains vs K x hipGraphLaunch of the captured chain (what concurrent batch
// solves would see).  Build: hipcc --offload-arch=gfx950 -O3 -w graph_mt_probe.hip -o graph_mt_probe -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_step(const double* __restrict__ in, double* __restrict__ out, int n, double a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a * in[i] + 1.0;
}
int main() {
  const int n = 100489, T = 512, B = (n + T - 1) / T, CHAIN = 12, REPS = 400;
  for (int K = 1; K <= 4; ++K) {
    for (int use_graph = 0; use_graph < 2; ++use_graph) {
      std::vector<std::thread> th;
      auto t0 = std::chrono::steady_clock::now();
      for (int t = 0; t < K; ++t)
        th.emplace_back([&, t] {
          (void)hipSetDevice(0);
          double *x, *y;
          (void)hipMalloc(&x, n * 8); (void)hipMalloc(&y, n * 8);
          hipStream_t s; (void)hipStreamCreate(&s);
          auto chain = [&]() { for (int k = 0; k < CHAIN; ++k) hipLaunchKernelGGL(k_step, dim3(B), dim3(T), 0, s, (k & 1) ? y : x, (k & 1) ? x : y, n, 0.5); };
          hipGraph_t g; hipGraphExec_t ge;
          (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
          chain();
          (void)hipStreamEndCapture(s, &g);
          (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
          for (int r = 0; r < REPS; ++r) {
            if (use_graph) (void)hipGraphLaunch(ge, s); else chain();
            while (hipStreamQuery(s) == hipErrorNotReady) {}
          }
          (void)hipStreamDestroy(s); (void)hipFree(x); (void)hipFree(y);
        });
      for (auto& t : th) t.join();
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      std::printf("threads %d %-22s: %.1f us per chain per thread, %.1f us per chain aggregate (includes ~setup)\n", K,
                  use_graph ? "hipGraphLaunch" : "12 x hipLaunchKernelGGL", us / REPS, us / REPS / K);
    }
  }
  return 0;
}
