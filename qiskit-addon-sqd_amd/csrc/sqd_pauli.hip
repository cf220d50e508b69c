// Projection of a Pauli-sum operator onto a subspace of computational basis states (qubit path).
//
// Replaces the per-term jax/numpy pipeline of the reference, qiskit_addon_sqd/qubit.py:
// matrix_elements_from_pauli (:167-240: XOR-connect via a bool "agreement map", sign/phase product,
// np.isin + np.searchsorted membership/lookup) and the term loop of project_operator_to_subspace
// (:78-144: operator += coefficient * coo_matrix(...), i.e. duplicates summed).
//
// Rows are the sorted unique integers of the bitstrings (column 0 = most significant bit, so the
// Pauli label character at position j acts on bit nbits-1-j).  For a term with masks (x, z):
//   connected state  = row ^ x
//   <conn| P |row>   = (-1)^{popcount(row & z)} * i^{popcount(x & z)}
// Terms that share the same x mask connect a row to the same column, so they are grouped: one binary
// search per (row, distinct x mask), one complex accumulation over the group's terms.  Output is CSR
// (row = input configuration, column = connected configuration -- the reference's convention),
// entries of a row in x-group order.
//
// gfx950 mapping: thread per row, rows coalesced; the sorted row table is the only irregularly
// accessed object and is L2 / Infinity-Cache resident up to ~3e7 rows; a diagonal group (x = 0) skips
// the search.  Two passes (count -> exclusive scan -> fill) give exact CSR without atomics.
#include <cstring>

#include "sqd_common.h"

namespace sqd {

__device__ inline int64_t lower_bound_u64(const uint64_t* __restrict__ a, int64_t n, uint64_t key) {
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void k_pauli_count(const uint64_t* __restrict__ rows, int64_t d, int ngroups,
                              const uint64_t* __restrict__ xmask, int64_t* __restrict__ cnt) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= d) return;
  const uint64_t row = rows[r];
  int64_t c = 0;
  for (int g = 0; g < ngroups; ++g) {
    const uint64_t x = xmask[g];
    if (x == 0) {
      ++c;
    } else {
      const uint64_t conn = row ^ x;
      const int64_t pos = lower_bound_u64(rows, d, conn);
      if (pos < d && rows[pos] == conn) ++c;
    }
  }
  cnt[r] = c;
}

__global__ void k_pauli_fill(const uint64_t* __restrict__ rows, int64_t d, int ngroups,
                             const uint64_t* __restrict__ xmask, const int64_t* __restrict__ group_ptr,
                             const uint64_t* __restrict__ zmask, const double* __restrict__ coef,
                             const int64_t* __restrict__ indptr, int64_t* __restrict__ indices,
                             double* __restrict__ data /* interleaved re, im */) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= d) return;
  const uint64_t row = rows[r];
  int64_t out = indptr[r];
  for (int g = 0; g < ngroups; ++g) {
    const uint64_t x = xmask[g];
    int64_t col = r;
    if (x != 0) {
      const uint64_t conn = row ^ x;
      col = lower_bound_u64(rows, d, conn);
      if (!(col < d && rows[col] == conn)) continue;
    }
    double re = 0.0, im = 0.0;
    for (int64_t t = group_ptr[g]; t < group_ptr[g + 1]; ++t) {
      const double s = (__popcll(row & zmask[t]) & 1) ? -1.0 : 1.0;
      re += s * coef[2 * t];
      im += s * coef[2 * t + 1];
    }
    indices[out] = col;
    data[2 * out] = re;
    data[2 * out + 1] = im;
    ++out;
  }
}

// exclusive scan by one workgroup (as in sqd_tables.hip; kept local so the qubit path is self-contained)
__global__ void k_pauli_scan(const int64_t* __restrict__ in, int64_t* __restrict__ out, int64_t n) {
  __shared__ int64_t sums[1024];
  const int T = blockDim.x, tid = threadIdx.x;
  const int64_t chunk = (n + T - 1) / T;
  const int64_t lo = (int64_t)tid * chunk;
  const int64_t hi = (lo + chunk < n) ? lo + chunk : n;
  int64_t s = 0;
  for (int64_t i = lo; i < hi; ++i) s += in[i];
  sums[tid] = s;
  __syncthreads();
  if (tid == 0) {
    int64_t run = 0;
    for (int t = 0; t < T; ++t) {
      const int64_t v = sums[t];
      sums[t] = run;
      run += v;
    }
    out[n] = run;
  }
  __syncthreads();
  int64_t run = sums[tid];
  for (int64_t i = lo; i < hi; ++i) {
    const int64_t v = in[i];
    out[i] = run;
    run += v;
  }
}

// ---- three-phase exclusive scan for large d: tile sums -> scan of tile sums (k_pauli_scan) -> tile scans
constexpr int SCAN_T = 256, SCAN_ITEMS = 16, SCAN_TILE = SCAN_T * SCAN_ITEMS;

__global__ void k_tile_sums(const int64_t* __restrict__ in, int64_t n, int64_t* __restrict__ tile_sum) {
  __shared__ int64_t red[SCAN_T];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
  int64_t s = 0;
  for (int k = 0; k < SCAN_ITEMS; ++k) {  // coalesced: consecutive lanes read consecutive elements
    const int64_t i = base + (int64_t)k * SCAN_T + threadIdx.x;
    if (i < n) s += in[i];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = SCAN_T / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = red[0];
}

// out[i] = tile_off[tile] + exclusive prefix inside the tile; thread t owns SCAN_ITEMS consecutive elements
__global__ void k_tile_scan(const int64_t* __restrict__ in, int64_t n, const int64_t* __restrict__ tile_off,
                            int64_t* __restrict__ out, int64_t ntiles) {
  __shared__ int64_t pre[SCAN_T];
  const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
  int64_t v[SCAN_ITEMS];
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    v[k] = (base + k < n) ? in[base + k] : 0;
    s += v[k];
  }
  pre[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t run = tile_off[blockIdx.x];
    for (int t = 0; t < SCAN_T; ++t) {
      const int64_t x = pre[t];
      pre[t] = run;
      run += x;
    }
    if (blockIdx.x == ntiles - 1) out[n] = run;
  }
  __syncthreads();
  int64_t run = pre[threadIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) out[base + k] = run;
    run += v[k];
  }
}

}  // namespace sqd

using namespace sqd;

struct sqd_pauli_plan {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  int64_t d = 0, nterms = 0, nnz = 0;
  int ngroups = 0;
  DevBuf rows, xmask, group_ptr, zmask, coef, cnt, indptr, indices, data, tiles;
  double ms_count = 0.0, ms_fill = 0.0;
};

#define SQD_API extern "C" __attribute__((visibility("default")))

SQD_API int sqd_pauli_free(sqd_pauli_plan* p) {
  if (!p) return SQD_OK;
  hipError_t e = hipSetDevice(p->device);
  (void)e;
  DevBuf* bufs[] = {&p->rows, &p->xmask, &p->group_ptr, &p->zmask, &p->coef, &p->cnt, &p->indptr, &p->indices, &p->data, &p->tiles};
  for (DevBuf* b : bufs) b->release();
  for (int i = 0; i < 2; ++i)
    if (p->ev[i]) e = hipEventDestroy(p->ev[i]);
  if (p->stream) e = hipStreamDestroy(p->stream);
  delete p;
  return SQD_OK;
}

SQD_API int sqd_pauli_count(int device, const uint64_t* rows, int64_t d, int ngroups, const uint64_t* xmask,
                            const int64_t* group_ptr, const uint64_t* zmask, const double* coef, int64_t* indptr_out,
                            int64_t* nnz_out, sqd_pauli_plan** plan_out) {
  if (!rows || d < 1 || ngroups < 1 || !xmask || !group_ptr || !zmask || !coef || !plan_out) {
    set_error("sqd_pauli_count: bad argument");
    return SQD_ERR_INVALID;
  }
  for (int64_t i = 1; i < d; ++i)
    if (!(rows[i - 1] < rows[i])) {
      set_error("sqd_pauli_count: rows must be strictly ascending (index " + std::to_string(i) + ")");
      return SQD_ERR_INVALID;
    }
  SQD_HIP_CHECK(hipSetDevice(device));
  sqd_pauli_plan* p = new sqd_pauli_plan();
  p->device = device;
  p->d = d;
  p->ngroups = ngroups;
  p->nterms = group_ptr[ngroups];
  int rc = SQD_OK;
  auto fail = [&](int code) {
    sqd_pauli_free(p);
    return code;
  };
  if (hipStreamCreate(&p->stream) != hipSuccess || hipEventCreate(&p->ev[0]) != hipSuccess ||
      hipEventCreate(&p->ev[1]) != hipSuccess) {
    set_error("sqd_pauli_count: stream/event creation failed");
    return fail(SQD_ERR_HIP);
  }
  struct Up { DevBuf* b; const void* src; size_t bytes; };
  const Up ups[] = {{&p->rows, rows, (size_t)d * 8}, {&p->xmask, xmask, (size_t)ngroups * 8},
                    {&p->group_ptr, group_ptr, (size_t)(ngroups + 1) * 8}, {&p->zmask, zmask, (size_t)p->nterms * 8},
                    {&p->coef, coef, (size_t)p->nterms * 16}};
  for (const Up& u : ups) {
    if ((rc = u.b->reserve(u.bytes + 8)) != SQD_OK) return fail(rc);
    if (hipMemcpyAsync(u.b->p, u.src, u.bytes, hipMemcpyHostToDevice, p->stream) != hipSuccess) {
      set_error("sqd_pauli_count: upload failed");
      return fail(SQD_ERR_HIP);
    }
  }
  if ((rc = p->cnt.reserve((size_t)d * 8)) != SQD_OK) return fail(rc);
  if ((rc = p->indptr.reserve((size_t)(d + 1) * 8)) != SQD_OK) return fail(rc);
  const unsigned nb = (unsigned)((d + 255) / 256);
  hipError_t e = hipEventRecord(p->ev[0], p->stream);
  hipLaunchKernelGGL(k_pauli_count, dim3(nb), dim3(256), 0, p->stream, (const uint64_t*)p->rows.as<uint64_t>(), d, ngroups,
                     (const uint64_t*)p->xmask.as<uint64_t>(), p->cnt.as<int64_t>());
  if (d <= 4 * SCAN_TILE) {
    hipLaunchKernelGGL(k_pauli_scan, dim3(1), dim3(256), 0, p->stream, (const int64_t*)p->cnt.as<int64_t>(),
                       p->indptr.as<int64_t>(), d);
  } else {
    const int64_t ntiles = (d + SCAN_TILE - 1) / SCAN_TILE;
    if ((rc = p->tiles.reserve((size_t)(2 * ntiles + 2) * 8)) != SQD_OK) return fail(rc);
    int64_t* tsum = p->tiles.as<int64_t>();
    int64_t* toff = tsum + ntiles;
    hipLaunchKernelGGL(k_tile_sums, dim3((unsigned)ntiles), dim3(SCAN_T), 0, p->stream, (const int64_t*)p->cnt.as<int64_t>(), d,
                       tsum);
    hipLaunchKernelGGL(k_pauli_scan, dim3(1), dim3(1024), 0, p->stream, (const int64_t*)tsum, toff, ntiles);
    hipLaunchKernelGGL(k_tile_scan, dim3((unsigned)ntiles), dim3(SCAN_T), 0, p->stream, (const int64_t*)p->cnt.as<int64_t>(),
                       d, (const int64_t*)toff, p->indptr.as<int64_t>(), ntiles);
  }
  e = hipEventRecord(p->ev[1], p->stream);
  if (hipGetLastError() != hipSuccess) {
    set_error("sqd_pauli_count: launch failed");
    return fail(SQD_ERR_HIP);
  }
  std::vector<int64_t> tmp;
  int64_t* dst = indptr_out;
  if (!dst) {
    tmp.resize(d + 1);
    dst = tmp.data();
  }
  e = hipMemcpyAsync(dst, p->indptr.p, (size_t)(d + 1) * 8, hipMemcpyDeviceToHost, p->stream);
  if (e != hipSuccess || hipStreamSynchronize(p->stream) != hipSuccess) {
    set_error(std::string("sqd_pauli_count: ") + hipGetErrorString(hipGetLastError()));
    return fail(SQD_ERR_HIP);
  }
  float ms = 0.f;
  e = hipEventElapsedTime(&ms, p->ev[0], p->ev[1]);
  p->ms_count = ms;
  p->nnz = dst[d];
  if (nnz_out) *nnz_out = p->nnz;
  *plan_out = p;
  return SQD_OK;
}

SQD_API int sqd_pauli_fill(sqd_pauli_plan* p, int64_t* indices, double* data, double* ms_kernels) {
  if (!p || !indices || !data) {
    set_error("sqd_pauli_fill: bad argument");
    return SQD_ERR_INVALID;
  }
  SQD_HIP_CHECK(hipSetDevice(p->device));
  SQD_TRY(p->indices.reserve((size_t)p->nnz * 8 + 8));
  SQD_TRY(p->data.reserve((size_t)p->nnz * 16 + 8));
  const unsigned nb = (unsigned)((p->d + 255) / 256);
  SQD_HIP_CHECK(hipEventRecord(p->ev[0], p->stream));
  hipLaunchKernelGGL(k_pauli_fill, dim3(nb), dim3(256), 0, p->stream, (const uint64_t*)p->rows.as<uint64_t>(), p->d,
                     p->ngroups, (const uint64_t*)p->xmask.as<uint64_t>(), (const int64_t*)p->group_ptr.as<int64_t>(),
                     (const uint64_t*)p->zmask.as<uint64_t>(), (const double*)p->coef.as<double>(),
                     (const int64_t*)p->indptr.as<int64_t>(), p->indices.as<int64_t>(), p->data.as<double>());
  SQD_HIP_CHECK(hipGetLastError());
  SQD_HIP_CHECK(hipEventRecord(p->ev[1], p->stream));
  if (p->nnz > 0) {
    SQD_HIP_CHECK(hipMemcpyAsync(indices, p->indices.p, (size_t)p->nnz * 8, hipMemcpyDeviceToHost, p->stream));
    SQD_HIP_CHECK(hipMemcpyAsync(data, p->data.p, (size_t)p->nnz * 16, hipMemcpyDeviceToHost, p->stream));
  }
  SQD_HIP_CHECK(hipStreamSynchronize(p->stream));
  float ms = 0.f;
  SQD_HIP_CHECK(hipEventElapsedTime(&ms, p->ev[0], p->ev[1]));
  p->ms_fill = ms;
  if (ms_kernels) *ms_kernels = p->ms_count + p->ms_fill;
  return SQD_OK;
}
