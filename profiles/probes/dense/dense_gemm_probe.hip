// Stand-alone probe (not product code): sigma += Ha * C + C * HbT with v_mfma_f64_16x16x4_f64, fragments
// loaded straight from global memory (L1/L2 resident at these sizes).  One wave per 32x32 tile of sigma.
// Build: hipcc --offload-arch=gfx950 -O3 -o dense_probe dense_gemm_probe.hip ; run: ./dense_probe 317 1000
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));

// D[M x N] (ld ldd) += A[M x K1] (lda) * B[K1 x N] (ldb)  +  A2[M x K2] * B2[K2 x N]
// all row-major, M, N, K padded to multiples of 32 / 4 by the caller (zero padded operands)
template <int WT>  // wave tile = WT x WT (16 or 32)
__global__ __launch_bounds__(64) void k_dense(int M, int N, int K1, const double* __restrict__ A, int lda,
                                              const double* __restrict__ B, int ldb, int K2,
                                              const double* __restrict__ A2, int lda2, const double* __restrict__ B2,
                                              int ldb2, double* __restrict__ D, int ldd) {
  constexpr int NT = WT / 16;
  const int lane = threadIdx.x;
  const int tn = N / WT;
  const int i0 = (blockIdx.x / tn) * WT, j0 = (blockIdx.x % tn) * WT;
  const int li = lane & 15, lk = lane >> 4;
  d4 acc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = d4{0, 0, 0, 0};
  for (int pass = 0; pass < 2; ++pass) {
    const double* __restrict__ Ap = pass ? A2 : A;
    const double* __restrict__ Bp = pass ? B2 : B;
    const int la = pass ? lda2 : lda, lb = pass ? ldb2 : ldb, K = pass ? K2 : K1;
    for (int k = 0; k < K; k += 4) {
      double af[NT], bf[NT];
#pragma unroll
      for (int a = 0; a < NT; ++a) af[a] = Ap[(size_t)(i0 + a * 16 + li) * la + k + lk];
#pragma unroll
      for (int b = 0; b < NT; ++b) bf[b] = Bp[(size_t)(k + lk) * lb + j0 + b * 16 + li];
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[a], bf[b], acc[a][b], 0, 0, 0);
    }
  }
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + a * 16 + lk + 4 * r, col = j0 + b * 16 + li;
        D[(size_t)row * ldd + col] += acc[a][b][r];
      }
}

int main(int argc, char** argv) {
  for (int ai = 1; ai < argc; ++ai) {
    const int n = atoi(argv[ai]);
    const int P = (n + 31) / 32 * 32;
    std::vector<double> Ha((size_t)P * P, 0), Hb((size_t)P * P, 0), C((size_t)P * P, 0), S((size_t)P * P, 0);
    srand(1);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) {
        Ha[(size_t)i * P + j] = (rand() % 4 == 0) ? rand() / (double)RAND_MAX - 0.5 : 0.0;
        Hb[(size_t)i * P + j] = (rand() % 4 == 0) ? rand() / (double)RAND_MAX - 0.5 : 0.0;
        C[(size_t)i * P + j] = rand() / (double)RAND_MAX - 0.5;
        S[(size_t)i * P + j] = i * 0.001 - j * 0.002;
      }
    double *dHa, *dHb, *dC, *dS;
    size_t bytes = (size_t)P * P * 8;
    hipMalloc(&dHa, bytes); hipMalloc(&dHb, bytes); hipMalloc(&dC, bytes); hipMalloc(&dS, bytes);
    hipMemcpy(dHa, Ha.data(), bytes, hipMemcpyHostToDevice);
    hipMemcpy(dHb, Hb.data(), bytes, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), bytes, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wt = 16; wt <= 32; wt += 16) {
      hipMemcpy(dS, S.data(), bytes, hipMemcpyHostToDevice);
      const int tiles = (P / wt) * (P / wt);
      auto launch = [&]() {
        if (wt == 16) hipLaunchKernelGGL(k_dense<16>, dim3(tiles), dim3(64), 0, 0, P, P, P, dHa, P, dC, P, P, dC, P, dHb, P, dS, P);
        else hipLaunchKernelGGL(k_dense<32>, dim3(tiles), dim3(64), 0, 0, P, P, P, dHa, P, dC, P, P, dC, P, dHb, P, dS, P);
      };
      launch();
      hipDeviceSynchronize();
      std::vector<double> out((size_t)P * P);
      hipMemcpy(out.data(), dS, bytes, hipMemcpyDeviceToHost);
      // check a sample of entries against the host
      double maxerr = 0;
      for (int t = 0; t < 2000; ++t) {
        int i = rand() % n, j = rand() % n;
        double ref = S[(size_t)i * P + j];
        for (int k = 0; k < n; ++k) ref += Ha[(size_t)i * P + k] * C[(size_t)k * P + j] + C[(size_t)i * P + k] * Hb[(size_t)k * P + j];
        maxerr = fmax(maxerr, fabs(ref - out[(size_t)i * P + j]));
      }
      const int reps = 20;
      hipEventRecord(e0);
      for (int r = 0; r < reps; ++r) launch();
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("n %d (pad %d) wave tile %d tiles %d: %.1f us per launch, maxerr %.2e, %.1f TFLOP/s\n", n, P, wt, tiles,
             ms * 1e3 / reps, maxerr, 4.0 * P * P * (double)P / (ms * 1e-3 / reps) / 1e12);
    }
    hipFree(dHa); hipFree(dHb); hipFree(dC); hipFree(dS);
  }
  return 0;
}
