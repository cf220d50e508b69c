// Small device-side helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>

namespace sqd {

// Sum over the workgroup; result valid on thread 0.  `red` = >= 16 doubles of LDS.
// Fixed shuffle tree + fixed wave order => bitwise reproducible.
__device__ inline double block_sum(double v, double* red) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) s += red[w];
  }
  return s;
}

}  // namespace sqd
