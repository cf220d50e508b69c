// GPU probe (not a test): does hipExtAnyOrderLaunch let two INDEPENDENT kernels of one stream overlap on gfx950?
// (hip_ext.h says the flag "is not supported on AMD GFX9xx boards" for hipExtModuleLaunchKernel.)
// Two kernels of 64 workgroups each spin for ~20 us; a third kernel follows in order.  Wall time per triple,
// with and without the flag on the second kernel, and each kernel's start/end clock.
//   hipcc --offload-arch=gfx950 -O2 anyorder_probe.hip -o anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
__global__ void k_spin(unsigned long long ticks, unsigned long long* stamp) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    atomicMin(&stamp[0], t0);
    atomicMax(&stamp[1], wall_clock64());
  }
}
int main() {
  unsigned long long* st;
  hipMalloc(&st, 6 * 8);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int flag = 0; flag < 2; ++flag) {
    double best = 1e9;
    unsigned long long h[6];
    for (int rep = 0; rep < 20; ++rep) {
      unsigned long long init[6] = {~0ull, 0, ~0ull, 0, ~0ull, 0};
      hipMemcpy(st, init, sizeof init, hipMemcpyHostToDevice);
      hipStreamSynchronize(s);
      auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, 2000ull, st);
      hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, flag ? hipExtAnyOrderLaunch : 0, 2000ull, st + 2);
      hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, 200ull, st + 4);
      hipStreamSynchronize(s);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (us < best) {
        best = us;
        hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
      }
    }
    const double t = (double)h[0];
    std::printf("flag %d: best wall %.1f us | k1 [%.1f, %.1f] k2 [%.1f, %.1f] k3 [%.1f, %.1f] us\n", flag, best, 0.0,
                (h[1] - t) / 100.0, (h[2] - t) / 100.0, (h[3] - t) / 100.0, (h[4] - t) / 100.0, (h[5] - t) / 100.0);
  }
  // fork / join over two streams with events: k0 (s) -> {k1 on s, k2 on s2} -> k3 (s)
  hipStream_t s2;
  hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  hipEvent_t e1, e2;
  hipEventCreateWithFlags(&e1, hipEventDisableTiming);
  hipEventCreateWithFlags(&e2, hipEventDisableTiming);
  unsigned long long* st2;
  hipMalloc(&st2, 8 * 8);
  for (int mode = 0; mode < 2; ++mode) {
    double best = 1e9;
    unsigned long long h[8];
    for (int rep = 0; rep < 30; ++rep) {
      unsigned long long init[8] = {~0ull, 0, ~0ull, 0, ~0ull, 0, ~0ull, 0};
      hipMemcpy(st2, init, sizeof init, hipMemcpyHostToDevice);
      hipStreamSynchronize(s);
      auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, 500ull, st2);
      if (mode == 1) {
        hipEventRecord(e1, s);
        hipStreamWaitEvent(s2, e1, 0);
      }
      hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, 2000ull, st2 + 2);
      hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, mode == 1 ? s2 : s, 2000ull, st2 + 4);
      if (mode == 1) {
        hipEventRecord(e2, s2);
        hipStreamWaitEvent(s, e2, 0);
      }
      hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, 200ull, st2 + 6);
      hipStreamSynchronize(s);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (us < best) {
        best = us;
        hipMemcpy(h, st2, sizeof h, hipMemcpyDeviceToHost);
      }
    }
    const double t = (double)h[0];
    std::printf("%s: best wall %.1f us | k0 [0, %.1f] k1 [%.1f, %.1f] k2 [%.1f, %.1f] k3 [%.1f, %.1f] us\n", mode ? "two streams + events" : "one stream",
                best, (h[1] - t) / 100.0, (h[2] - t) / 100.0, (h[3] - t) / 100.0, (h[4] - t) / 100.0, (h[5] - t) / 100.0,
                (h[6] - t) / 100.0, (h[7] - t) / 100.0);
  }
  return 0;
}
