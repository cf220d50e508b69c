// Stand-alone probe (not product code), second version: sigma += Ha * C + C * Hb^T on the f64 matrix cores
// (v_mfma_f64_16x16x4_f64) the way it would have to be built to compete with the sparse same-spin work items:
//  * Ha, Hb are SYMMETRIC, so every "A[i][k]" fragment of them is read as [k][i]: coalesced (4 rows x 128 B per load);
//  * the one operand that is not symmetric -- the rows of C in the second product -- is staged through LDS, coalesced
//    on the way in, padded pitch on the way out;
//  * split-K: the 4 wavefronts of a workgroup share one 32 x 32 tile of sigma, each taking a quarter of the k range of
//    both products; the four partial tiles meet in LDS and are added in wave order (fixed => reproducible);
//  * 8 k-steps of fragments requested before the 8 x 4 MFMAs that consume them.
// Build: hipcc --offload-arch=gfx950 -O3 -o dense_probe2 dense_gemm_probe2.hip ; run: ./dense_probe2 317 707 1000
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int WT = 32, NT = WT / 16, KS = 4 /* waves = k splits */, KC = 32 /* k chunk staged for C rows */, PITCH = KC + 1;

// D[P x P] += Ha[P x P] * C[P x P] + C[P x P] * Hb[P x P]   (Ha, Hb symmetric; all row-major, ld = P, P % 64 == 0)
__global__ __launch_bounds__(64 * KS) void k_dense2(int P, const double* __restrict__ Ha, const double* __restrict__ Hb,
                                                    const double* __restrict__ C, double* __restrict__ D) {
  __shared__ double s_c[KS][WT * PITCH];       // per wave: its chunk of the C rows of the tile
  __shared__ double s_red[KS][WT * WT];        // partial tiles
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int tn = P / WT;
  const int i0 = (blockIdx.x / tn) * WT, j0 = (blockIdx.x % tn) * WT;
  const int li = lane & 15, lk = lane >> 4;
  const int kq = P / KS, k_lo = wv * kq, k_hi = k_lo + kq;  // this wave's k range (P % (4 KS) == 0)
  d4 acc[NT][NT];
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b) acc[a][b] = d4{0, 0, 0, 0};
  // ---- product 1: Ha * C, A fragment Ha[i][k] read as Ha[k][i]
  for (int k = k_lo; k < k_hi; k += 32) {
    double af[8][NT], bf[8][NT];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int a = 0; a < NT; ++a) af[u][a] = Ha[(size_t)(k + 4 * u + lk) * P + i0 + a * 16 + li];
#pragma unroll
      for (int b = 0; b < NT; ++b) bf[u][b] = C[(size_t)(k + 4 * u + lk) * P + j0 + b * 16 + li];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][a], bf[u][b], acc[a][b], 0, 0, 0);
  }
  // ---- product 2: C * Hb^T = C * Hb; A fragment C[i][k] through LDS, B fragment Hb[k][j] coalesced
  double* sc = s_c[wv];
  for (int kc = k_lo; kc < k_hi; kc += KC) {
    // stage C[i0 .. i0+32)[kc .. kc+KC): one row segment (512 B) per 64 lanes
#pragma unroll 4
    for (int r = 0; r < WT; ++r) if (lane < KC) sc[r * PITCH + lane] = C[(size_t)(i0 + r) * P + kc + lane];
    __builtin_amdgcn_wave_barrier();
    for (int k = 0; k < KC; k += 32) {
      double af[8][NT], bf[8][NT];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
#pragma unroll
        for (int a = 0; a < NT; ++a) af[u][a] = sc[(a * 16 + li) * PITCH + k + 4 * u + lk];
#pragma unroll
        for (int b = 0; b < NT; ++b) bf[u][b] = Hb[(size_t)(kc + k + 4 * u + lk) * P + j0 + b * 16 + li];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
          for (int b = 0; b < NT; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][a], bf[u][b], acc[a][b], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  // ---- the KS partial tiles meet in LDS, added in wave order
#pragma unroll
  for (int a = 0; a < NT; ++a)
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) s_red[wv][(a * 16 + lk + 4 * r) * WT + b * 16 + li] = acc[a][b][r];
  __syncthreads();
  for (int e = threadIdx.x; e < WT * WT; e += blockDim.x) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < KS; ++w) s += s_red[w][e];
    D[(size_t)(i0 + e / WT) * P + j0 + e % WT] += s;
  }
}

int main(int argc, char** argv) {
  for (int ai = 1; ai < argc; ++ai) {
    const int n = atoi(argv[ai]);
    const int P = (n + 127) / 128 * 128;  // k range per wave a multiple of KC = 32
    std::vector<double> Ha((size_t)P * P, 0), Hb((size_t)P * P, 0), C((size_t)P * P, 0), S((size_t)P * P, 0);
    srand(1);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j <= i; ++j) {
        const double a = (rand() % 4 == 0) ? rand() / (double)RAND_MAX - 0.5 : 0.0, b = (rand() % 4 == 0) ? rand() / (double)RAND_MAX - 0.5 : 0.0;
        Ha[(size_t)i * P + j] = Ha[(size_t)j * P + i] = a;
        Hb[(size_t)i * P + j] = Hb[(size_t)j * P + i] = b;
      }
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) { C[(size_t)i * P + j] = rand() / (double)RAND_MAX - 0.5; S[(size_t)i * P + j] = i * 0.001 - j * 0.002; }
    double *dHa, *dHb, *dC, *dS;
    size_t bytes = (size_t)P * P * 8;
    hipMalloc(&dHa, bytes); hipMalloc(&dHb, bytes); hipMalloc(&dC, bytes); hipMalloc(&dS, bytes);
    hipMemcpy(dHa, Ha.data(), bytes, hipMemcpyHostToDevice);
    hipMemcpy(dHb, Hb.data(), bytes, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), bytes, hipMemcpyHostToDevice);
    hipMemcpy(dS, S.data(), bytes, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int tiles = (P / WT) * (P / WT);
    auto launch = [&]() { hipLaunchKernelGGL(k_dense2, dim3(tiles), dim3(64 * KS), 0, 0, P, dHa, dHb, dC, dS); };
    launch();
    hipDeviceSynchronize();
    std::vector<double> out((size_t)P * P);
    hipMemcpy(out.data(), dS, bytes, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int t = 0; t < 2000; ++t) {
      int i = rand() % n, j = rand() % n;
      double ref = S[(size_t)i * P + j];
      for (int k = 0; k < n; ++k) ref += Ha[(size_t)i * P + k] * C[(size_t)k * P + j] + C[(size_t)i * P + k] * Hb[(size_t)k * P + j];
      maxerr = fmax(maxerr, fabs(ref - out[(size_t)i * P + j]));
    }
    const int reps = 50;
    for (int r = 0; r < 5; ++r) launch();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("n %d (pad %d) 32x32 tiles %d x %d waves (split-K): %.1f us per launch, maxerr %.2e, %.1f TFLOP/s on the padded size, %.1f on n\n",
           n, P, tiles, KS, ms * 1e3 / reps, maxerr, 4.0 * P * P * (double)P / (ms * 1e-3 / reps) / 1e12,
           4.0 * n * (double)n * n / (ms * 1e-3 / reps) / 1e12);
    hipFree(dHa); hipFree(dHb); hipFree(dC); hipFree(dS);
  }
  return 0;
}
