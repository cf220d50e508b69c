// GPU probe (not product code): what a hipGraph buys for a chain of DEPENDENT small kernels like one Davidson round
// (4 launches of ~200 workgroups x 512 threads, each touching 0.8 MB).  Compares, for a chain of 12 kernels:
//   (a) 12 hipLaunchKernelGGL calls on one stream, (b) one hipGraphLaunch of the captured chain.
// Build: hipcc --offload-arch=gfx950 -O3 graph_probe.hip -o graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_step(const double* __restrict__ in, double* __restrict__ out, int n, double a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a * in[i] + 1.0;
}
int main() {
  const int n = 100489, T = 512, B = (n + T - 1) / T, CHAIN = 12, REPS = 300;
  double *x, *y;
  CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, n * 8));
  CK(hipMemset(x, 0, n * 8)); CK(hipMemset(y, 0, n * 8));
  hipStream_t s; CK(hipStreamCreate(&s));
  auto chain = [&]() { for (int k = 0; k < CHAIN; ++k) hipLaunchKernelGGL(k_step, dim3(B), dim3(T), 0, s, (k & 1) ? y : x, (k & 1) ? x : y, n, 0.5); };
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  chain();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto med = [&](auto f) {
    std::vector<double> t;
    for (int r = 0; r < REPS + 50; ++r) {
      auto t0 = now(); f();
      while (hipStreamQuery(s) == hipErrorNotReady) {}
      auto t1 = now();
      if (r >= 50) t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
  };
  const double a = med([&] { chain(); });
  const double b = med([&] { (void)hipGraphLaunch(ge, s); });
  // device-side span of the chain (events around it), both ways
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto span = [&](auto f) {
    std::vector<double> t;
    for (int r = 0; r < 100; ++r) {
      (void)hipEventRecord(e0, s); f(); (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); t.push_back(ms * 1e3);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
  };
  const double sa = span([&] { chain(); }), sb = span([&] { (void)hipGraphLaunch(ge, s); });
  std::printf("chain of %d dependent kernels (%d blocks x %d threads): host-visible completion  stream launches %.1f us | hipGraphLaunch %.1f us\n", CHAIN, B, T, a, b);
  std::printf("                                                      device span (events)     stream launches %.1f us | hipGraphLaunch %.1f us\n", sa, sb);
  std::printf("per kernel: %.2f / %.2f us (host-visible), %.2f / %.2f us (device span)\n", a / CHAIN, b / CHAIN, sa / CHAIN, sb / CHAIN);
  return 0;
}
