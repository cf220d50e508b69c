// Reduced density matrices of a subspace state from the same link tables the sigma kernel uses.
//
// Replaces pyscf SelectedCI.make_rdm1s / make_rdm1 / make_rdm2 (FCImake_rdm1a/b, FCItdm12kern_ab,
// SCIrdm2_aaaa), which the reference calls at qiskit_addon_sqd/fermion.py:725-729, :821-826 and
// :117-125 -- each of those calls rebuilds pyscf's link tables; here they are already resident.
//
// Conventions: dm1s[p,q] = <a+_p a_q>;  dm2[p,q,r,s] = sum_{st} <p+_s r+_t s_t q_s>.
// Orbital occupancies (the diagonal of dm1s, the only part that feeds back into the SQD loop) are
// reduced in a fixed order and are bitwise reproducible; off-diagonal bins use f64 atomics.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sqd_common.h"
#include "sqd_device.h"
#include "sqd_direct.h"
#include "sqd_davstate.h"

namespace sqd {

// w[A] = sum_B C[A,B]^2   (one wavefront per row)
__global__ void k_row_norms(const double* __restrict__ C, int64_t na, int64_t nb, double* __restrict__ w) {
  const int lane = threadIdx.x & 63;
  const int64_t A = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (A >= na) return;
  double s = 0.0;
  for (int64_t b = lane; b < nb; b += 64) {
    const double v = C[A * nb + b];
    s += v * v;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (lane == 0) w[A] = s;
}
// w[B] = sum_A C[A,B]^2.  Workgroup = 64 columns (unit stride across a wave) x RL row lanes that
// split the A loop; partials meet in LDS and are added in fixed order.
__global__ void k_col_norms(const double* __restrict__ C, int64_t na, int64_t nb, double* __restrict__ w) {
  __shared__ double red[1024];
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6, RL = blockDim.x >> 6;
  const int64_t B = (int64_t)blockIdx.x * 64 + col;
  double s = 0.0;
  if (B < nb)
    for (int64_t a0 = rl; a0 < na; a0 += (int64_t)RL * 8) {
      // eight rows requested together (rows past the end re-read row a0 and are not added): as a plain loop
      // this was one memory round trip per row, 12 us for 317 rows
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t a = a0 + (int64_t)u * RL;
        v[u] = C[(a < na ? a : a0) * nb + B];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (a0 + (int64_t)u * RL < na) ? v[u] * v[u] : 0.0;
    }
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && B < nb) {
    for (int r = 1; r < RL; ++r) s += red[r * 64 + col];
    w[B] = s;
  }
}
// out[l] = sum_B C[tgt_l,B] C[src_l,B]   (one wavefront per alpha link)
__global__ void k_rowpair_dots(const double* __restrict__ C, int64_t nb, int64_t nl, const uint32_t* __restrict__ tgt,
                               const uint32_t* __restrict__ src, int src_stride, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t l = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (l >= nl) return;
  const double* r0 = C + (int64_t)tgt[l] * nb;
  const double* r1 = C + (int64_t)src[(int64_t)l * src_stride] * nb;
  double s = 0.0;
  for (int64_t b = lane; b < nb; b += 64) s += r0[b] * r1[b];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  if (lane == 0) out[l] = s;
}
// out[l] = sum_A C[A,tgt_l] C[A,src_l]   (thread per beta link)
__global__ void k_colpair_dots(const double* __restrict__ C, int64_t na, int64_t nb, int64_t nl,
                               const uint32_t* __restrict__ tgt, const uint32_t* __restrict__ src, int src_stride,
                               double* __restrict__ out) {
  __shared__ double red[1024];
  const int col = threadIdx.x & 63, rl = threadIdx.x >> 6, RL = blockDim.x >> 6;
  const int64_t l = (int64_t)blockIdx.x * 64 + col;
  double s = 0.0;
  if (l < nl) {
    const int64_t b0 = tgt[l], b1 = src[(int64_t)l * src_stride];
    for (int64_t a = rl; a < na; a += RL) s += C[a * nb + b0] * C[a * nb + b1];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (rl == 0 && l < nl) {
    for (int r = 1; r < RL; ++r) s += red[r * 64 + col];
    out[l] = s;
  }
}

// occ[p] = sum_I [p in I] w[I]: one workgroup per orbital, fixed partition + fixed tree
__global__ void k_occupancy(const uint64_t* __restrict__ strs, int64_t n, const double* __restrict__ w, int stride,
                            double* __restrict__ out /* out[p*stride] */) {
  __shared__ double red[16];
  const int p = blockIdx.x;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += blockDim.x)
    if ((strs[i] >> p) & 1ull) s += w[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) out[(int64_t)p * stride] = s;
}

__global__ void k_rdm1_singles(int64_t nl, const SRec* __restrict__ rec, const double* __restrict__ dots, int norb,
                               double* __restrict__ dm1) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= nl) return;
  const uint32_t m = rec[l].meta;
  atomicAdd(&dm1[srec_cre(m) * norb + srec_des(m)], srec_sign(m) * dots[l]);
}

// ---- dm2 same-spin pieces (operator index order [p,q,r,s] <-> a+_p a+_r a_s a_q)
__device__ inline int64_t i4(int n, int p, int q, int r, int s) { return (((int64_t)p * n + q) * n + r) * n + s; }

__global__ void k_rdm2_diag(const uint64_t* __restrict__ strs, int64_t n, const double* __restrict__ w, int norb,
                            double* __restrict__ dm2) {
  // thread per (p,r) pair, fixed-order loop over strings
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= norb * norb) return;
  const int p = idx / norb, r = idx % norb;
  if (p == r) return;
  const uint64_t mask = (1ull << p) | (1ull << r);
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i)
    if ((strs[i] & mask) == mask) s += w[i];
  atomicAdd(&dm2[i4(norb, p, p, r, r)], s);
  atomicAdd(&dm2[i4(norb, p, r, r, p)], -s);
}
__global__ void k_rdm2_singles(const uint64_t* __restrict__ strs, int64_t nl, const SRec* __restrict__ rec,
                               const double* __restrict__ dots, int norb, double* __restrict__ dm2) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= nl) return;
  const uint32_t m = rec[l].meta;
  const int a = srec_cre(m), b = srec_des(m);
  const double ov = srec_sign(m) * dots[l];
  uint64_t occ = strs[rec[l].src] & ~(1ull << b);
  while (occ) {
    const int k = __ffsll((long long)occ) - 1;
    occ &= occ - 1;
    atomicAdd(&dm2[i4(norb, a, b, k, k)], ov);
    atomicAdd(&dm2[i4(norb, k, k, a, b)], ov);
    atomicAdd(&dm2[i4(norb, a, k, k, b)], -ov);
    atomicAdd(&dm2[i4(norb, k, b, a, k)], -ov);
  }
}
__global__ void k_rdm2_doubles(int64_t nl, const uint32_t* __restrict__ orb, const double* __restrict__ dots, int norb,
                               double* __restrict__ dm2) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= nl) return;
  const uint32_t o = orb[l];
  const int p = o & 63, r = (o >> 6) & 63, q = (o >> 12) & 63, s = (o >> 18) & 63;
  const double ov = ((o >> 31) ? -1.0 : 1.0) * dots[l];
  atomicAdd(&dm2[i4(norb, p, q, r, s)], ov);
  atomicAdd(&dm2[i4(norb, r, s, p, q)], ov);
  atomicAdd(&dm2[i4(norb, p, s, r, q)], -ov);
  atomicAdd(&dm2[i4(norb, r, q, p, s)], -ov);
}

// ---- dm2 opposite-spin: extended link lists (singles + one diagonal pseudo-link per occupied orbital)
struct XLink {
  uint32_t tgt, src;
  uint32_t pq;     // cre | des<<6
  float sign;
};
__global__ void k_xlinks(const uint64_t* __restrict__ strs, int64_t n, int nocc, int64_t n_s,
                         const uint32_t* __restrict__ s_row, const SRec* __restrict__ s_rec, XLink* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_s) {
    const uint32_t m = s_rec[i].meta;
    out[i] = XLink{s_row[i], s_rec[i].src, srec_cre(m) | (srec_des(m) << 6), (float)srec_sign(m)};
  } else if (i < n_s + n * nocc) {
    const int64_t j = i - n_s;
    const int64_t I = j / nocc;
    int k = (int)(j % nocc);
    uint64_t occ = strs[I];
    for (int t = 0; t < k; ++t) occ &= occ - 1;
    const uint32_t p = (uint32_t)(__ffsll((long long)occ) - 1);
    out[i] = XLink{(uint32_t)I, (uint32_t)I, p | (p << 6), 1.0f};
  }
}
// G[p,q,r,s] += sa sb C[tgt_a,tgt_b] C[src_a,src_b]  over all (alpha xlink, beta xlink) pairs
__global__ void k_rdm2_ab(const double* __restrict__ C, int64_t nb, int64_t la, int64_t lb,
                          const XLink* __restrict__ xa, const XLink* __restrict__ xb, int norb,
                          double* __restrict__ G) {
  const int64_t ib = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ib >= lb) return;
  const XLink b = xb[ib];
  const int r = b.pq & 63, s = (b.pq >> 6) & 63;
  for (int64_t ia = blockIdx.y; ia < la; ia += gridDim.y) {
    const XLink a = xa[ia];
    const int p = a.pq & 63, q = (a.pq >> 6) & 63;
    const double v = (double)(a.sign * b.sign) * C[(int64_t)a.tgt * nb + b.tgt] * C[(int64_t)a.src * nb + b.src];
    if (v != 0.0) atomicAdd(&G[i4(norb, p, q, r, s)], v);
  }
}
// ---- the same sum for CONNECTED sets (la x lb ~ 3e9 pairs at 3000 strings per spin: 132 ms of one global atomic per pair
// above).  The pairs factorise like the single x single term of sigma (sqd_opp.hip): for an alpha xlink a, every beta xlink
// b contributes sa sb u[tgt_b] v[src_b] with u = C[tgt_a, :], v = C[src_a, :] to G[pq_a][rs_b].  Both lists are sorted by
// their orbital pair on the device (counting sort: k_xl_hist / k_xl_scan / k_xl_scatter); a workgroup owns a chunk of <= E
// alpha xlinks of ONE pair pq and a range of the beta list: every thread keeps S consecutive beta xlinks (packed, in
// registers) and one sum per xlink over the whole chunk, the two rows are staged in LDS per alpha xlink; at the end the
// sums fold over the runs of equal rs inside the thread and go to G[pq][rs] with one atomic per run.
constexpr int RAB_S = 12, RAB_E = 32;
__global__ void k_xl_hist(const XLink* __restrict__ x, int64_t n, unsigned* __restrict__ hist) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&hist[x[i].pq & 4095u], 1u);
}
__global__ void __launch_bounds__(1024) k_xl_scan(const unsigned* __restrict__ hist, unsigned* __restrict__ cursor) {  // 4096 bins, one workgroup
  __shared__ unsigned part[1024];
  const int t = threadIdx.x;
  unsigned v[4], sum = 0;
  for (int k = 0; k < 4; ++k) v[k] = hist[4 * t + k], sum += v[k];
  part[t] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const unsigned y = t >= d ? part[t - d] : 0u;
    __syncthreads();
    part[t] += y;
    __syncthreads();
  }
  unsigned run = part[t] - sum;
  for (int k = 0; k < 4; ++k) cursor[4 * t + k] = run, run += v[k];
}
__global__ void k_xl_scatter(const XLink* __restrict__ x, int64_t n, unsigned* __restrict__ cursor, XLink* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[atomicAdd(&cursor[x[i].pq & 4095u], 1u)] = x[i];
}
struct RabItem {
  uint32_t a0, na;  // alpha xlinks [a0, a0 + na) of the sorted list: one orbital pair
};
struct RabArgs {
  GPtr<const double> c;
  int64_t nb, lb;
  GPtr<const XLink> xa, xb;  // sorted by pq
  GPtr<const RabItem> items;
  unsigned n_items;
  int H, T, norb;
  GPtr<double> G;
};
__global__ void __launch_bounds__(1024) k_rdm2_ab_rows(const RabArgs g) {
  HIP_DYNAMIC_SHARED(double, smem)  // u[nb] | v[nb]
  const int T = g.T, tid = threadIdx.x;
  const unsigned item = blockIdx.x / (unsigned)g.H;
  const int h = (int)(blockIdx.x % (unsigned)g.H);
  if (item >= g.n_items) return;
  const RabItem it = g.items[item];
  const int64_t nb = g.nb;
  double* u = smem;
  double* v = smem + ((nb + 1) & ~int64_t(1));
  // this thread's beta xlinks: {tgt | src << 16}, {rs | live << 30 | negative << 31}
  uint32_t w0[RAB_S], w1[RAB_S];
  double acc[RAB_S];
  const int64_t l0 = ((int64_t)h * T + tid) * RAB_S;
#pragma unroll
  for (int s = 0; s < RAB_S; ++s) {
    const int64_t l = l0 + s;
    w0[s] = 0u, w1[s] = 0u, acc[s] = 0.0;
    if (l < g.lb) {
      const XLink b = g.xb[l];
      w0[s] = (b.tgt & 0xffffu) | (b.src << 16);
      w1[s] = (b.pq & 4095u) | (1u << 30) | (b.sign < 0.0f ? (1u << 31) : 0u);
    }
  }
  for (uint32_t k = 0; k < it.na; ++k) {
    const XLink a = g.xa[it.a0 + k];
    const double* __restrict__ ru = g.c + (int64_t)a.tgt * nb;
    const double* __restrict__ rv = g.c + (int64_t)a.src * nb;
    const double sa = (double)a.sign;
    __syncthreads();  // (the previous pair of rows has been read)
    for (int64_t B = tid; B < nb; B += T) {
      u[B] = sa * ru[B];
      v[B] = rv[B];
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < RAB_S; ++s) acc[s] += u[w0[s] & 0xffffu] * v[w0[s] >> 16];
  }
  // runs of equal rs inside the thread -> one atomic each (dead slots carry no live bit)
  const XLink a0 = g.xa[it.a0];
  const int p = a0.pq & 63, q = (a0.pq >> 6) & 63;
  double run = 0.0;
#pragma unroll
  for (int s = 0; s < RAB_S; ++s) {
    if (w1[s] & (1u << 30)) {
      run += (w1[s] >> 31) ? -acc[s] : acc[s];
      const bool last = (s == RAB_S - 1) || !(w1[s + 1 < RAB_S ? s + 1 : s] & (1u << 30)) ||
                        ((w1[s + 1 < RAB_S ? s + 1 : s] ^ w1[s]) & 4095u);
      if (last) {
        const int r = w1[s] & 63, sq = (w1[s] >> 6) & 63;
        if (run != 0.0) atomicAdd(&g.G[i4(g.norb, p, q, r, sq)], run);
        run = 0.0;
      }
    }
  }
}
// dm2 += G + G^T(2,3,0,1)
__global__ void k_rdm2_symm_add(int norb, const double* __restrict__ G, double* __restrict__ dm2) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n2 = (int64_t)norb * norb;
  if (idx >= n2 * n2) return;
  const int64_t pq = idx / n2, rs = idx % n2;
  dm2[idx] += G[idx] + G[rs * n2 + pq];
}

static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

// link dots for one spin into `dots` (singles first, then doubles); norms into w
static int spin_link_dots(sqd_ctx* c, int spin, const double* C, double* w, double* dots_s, double* dots_d) {
  const SpinTables& t = c->sp[spin];
  hipStream_t st = c->stream;
  if (spin == 0) {
    hipLaunchKernelGGL(k_row_norms, dim3(nblk(t.n, 4)), dim3(256), 0, st, C, c->na, c->nb, w);
    if (t.n_s > 0)
      hipLaunchKernelGGL(k_rowpair_dots, dim3(nblk(t.n_s, 4)), dim3(256), 0, st, C, c->nb, t.n_s,
                         (const uint32_t*)t.s_row.as<uint32_t>(), (const uint32_t*)t.s_rec.as<uint32_t>(), 2, dots_s);
    if (t.n_d > 0 && dots_d)
      hipLaunchKernelGGL(k_rowpair_dots, dim3(nblk(t.n_d, 4)), dim3(256), 0, st, C, c->nb, t.n_d,
                         (const uint32_t*)t.d_row.as<uint32_t>(), (const uint32_t*)t.d_src.as<uint32_t>(), 1, dots_d);
  } else {
    hipLaunchKernelGGL(k_col_norms, dim3(nblk(t.n, 64)), dim3(512), 0, st, C, c->na, c->nb, w);
    if (t.n_s > 0)
      hipLaunchKernelGGL(k_colpair_dots, dim3(nblk(t.n_s, 64)), dim3(512), 0, st, C, c->na, c->nb, t.n_s,
                         (const uint32_t*)t.s_row.as<uint32_t>(), (const uint32_t*)t.s_rec.as<uint32_t>(), 2, dots_s);
    if (t.n_d > 0 && dots_d)
      hipLaunchKernelGGL(k_colpair_dots, dim3(nblk(t.n_d, 64)), dim3(512), 0, st, C, c->na, c->nb, t.n_d,
                         (const uint32_t*)t.d_row.as<uint32_t>(), (const uint32_t*)t.d_src.as<uint32_t>(), 1, dots_d);
  }
  SQD_HIP_CHECK(hipGetLastError());
  return SQD_OK;
}

int dev_rdm1s(sqd_ctx* c, const double* d_c, double* dm1a, double* dm1b) {
  const int norb = c->norb;
  const int64_t n2 = (int64_t)norb * norb;
  hipStream_t st = c->stream;
  int64_t need = 2 * n2;
  for (int s = 0; s < 2; ++s) need += c->sp[s].n + c->sp[s].n_s;
  SQD_TRY(c->scratch.reserve((size_t)need * 8 + 64));
  double* base = c->scratch.as<double>();
  double* dm = base;  // [2][n2]
  SQD_HIP_CHECK(hipMemsetAsync(dm, 0, 2 * n2 * 8, st));
  double* p = base + 2 * n2;
  for (int s = 0; s < 2; ++s) {
    const SpinTables& t = c->sp[s];
    double* w = p;
    double* dots = p + t.n;
    p += t.n + t.n_s;
    SQD_TRY(spin_link_dots(c, s, d_c, w, dots, nullptr));
    hipLaunchKernelGGL(k_occupancy, dim3(norb), dim3(256), 0, st, (const uint64_t*)t.strs.as<uint64_t>(), t.n,
                       (const double*)w, norb + 1, dm + s * n2);
    if (t.n_s > 0)
      hipLaunchKernelGGL(k_rdm1_singles, dim3(nblk(t.n_s, 256)), dim3(256), 0, st, t.n_s,
                         (const SRec*)t.s_rec.as<SRec>(), (const double*)dots, norb, dm + s * n2);
    SQD_HIP_CHECK(hipGetLastError());
  }
  SQD_HIP_CHECK(hipMemcpyAsync(dm1a, dm, n2 * 8, hipMemcpyDeviceToHost, st));
  SQD_HIP_CHECK(hipMemcpyAsync(dm1b, dm + n2, n2 * 8, hipMemcpyDeviceToHost, st));
  SQD_STREAM_SYNC(st);
  return SQD_OK;
}

// ---- everything solve_fermion derives from the state after the Davidson (reference fermion.py:820-830) in ONE
// launch: the dot products c.(Hc), c.(S^2 c), c.c, |S^2 c|^2 and both spins' orbital occupancies (the diagonals of
// pyscf make_rdm1s).  Two kinds of workgroup in the same grid:
//   row role    (blockIdx.x <  nrb): one wavefront per alpha string A -- w_a[A] = sum_B C[A,B]^2 and the dot products
//                over its row; the workgroup adds its rows' weights into a partial occ_a[p] (fixed order);
//   column role (blockIdx.x >= nrb): 64 beta strings x 8 row lanes -- w_b[B] = sum_A C[A,B]^2 (eight rows in flight
//                per lane), then a partial occ_b[p] over its 64 strings.
// The workgroup that arrives last folds the partial records in block order (bitwise reproducible) and writes the
// result straight into host-visible memory: no reduce launches, no copy.
constexpr int OBS_W = 4 + SQD_MAX_NORB;  // partial record: 4 dot products + one occupancy per orbital
constexpr int OBS_ROWS = 8;              // alpha strings per row-role workgroup (512 threads)
struct ObsArgs {
  GPtr<const double> C, T1, T2;
  int64_t na, nb;
  GPtr<const uint64_t> strs_a, strs_b;
  int norb;
  unsigned nrb, gx;  // row-role workgroups, all workgroups of this subspace
  GPtr<double> partial;
  GPtr<unsigned> counter;
  GPtr<double> out;
  GPtr<double> host_c;
  GPtr<long long> seq_word;
  long long seq;

  int s2_inline;
  DirectArgs dg;
  // optional device-side copy of the results, led by the Davidson eigenvalue (sqd_ctx::record_out): what a collective
  // exchange on the same stream reads
  GPtr<double> record;
  GPtr<const DavState> st;
};
__device__ inline void observables_finish(const ObsArgs& g, unsigned nbx, double* red);
__device__ inline void observables_body(const ObsArgs& g, unsigned bx, unsigned nbx) {
  __shared__ double red[1024];
  __shared__ double wrow[64];
  __shared__ double dots[OBS_ROWS][4];
  const double* __restrict__ C = g.C;
  const double* __restrict__ T1 = g.T1;
  const double* __restrict__ T2 = g.T2;
  const int64_t na = g.na, nb = g.nb;
  const uint64_t* __restrict__ strs_a = g.strs_a;
  const uint64_t* __restrict__ strs_b = g.strs_b;
  const int norb = g.norb;
  const unsigned nrb = g.nrb;
  double* partial = g.partial;
  double* out = g.out;
  double* __restrict__ host_c = g.host_c;
  const int s2_inline = g.s2_inline;
  const DirectArgs& dg = g.dg;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double* rec = partial + (int64_t)bx * OBS_W;
  if (bx < nrb) {
    const int64_t A = (int64_t)bx * OBS_ROWS + wv;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    if (A < na) {
      // eight elements per lane requested per round (a row of the headline problem is five: one round trip instead
      // of five in sequence), accumulated in the order of the plain loop.  host_c != nullptr: this pass is also the
      // transfer of the state to the caller's page-locked buffer -- full-line posted writes over PCIe from the
      // kernel that reads every element anyway, instead of a DMA copy on a second stream behind an event.
      for (int64_t b0 = lane; b0 < nb; b0 += 64 * 8) {
        double v[8], h[8], t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t b = b0 + 64 * u, idx = A * nb + (b < nb ? b : b0);
          v[u] = C[idx];
          h[u] = T1 ? T1[idx] : 0.0;
          // (s2_inline: S^2 c of this element evaluated here from the CSR lists -- ultra-sparse sets, where a
          // sigma launch of its own would cost more than the handful of links it walks)
          t[u] = s2_inline ? direct_element<true>(dg, C, idx, -1.0) : (T2 ? T2[idx] : 0.0);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool ok = b0 + 64 * u < nb;
          if (ok && host_c) __builtin_nontemporal_store(v[u], &host_c[A * nb + b0 + 64 * u]);
          s[0] += ok ? h[u] * v[u] : 0.0;
          s[1] += ok ? t[u] * v[u] : 0.0;
          s[2] += ok ? v[u] * v[u] : 0.0;
          s[3] += ok ? t[u] * t[u] : 0.0;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      for (int off = 32; off > 0; off >>= 1) s[k] += __shfl_down(s[k], off);
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) dots[wv][k] = s[k];
      wrow[wv] = s[2];
    }
    __syncthreads();
    if ((int)threadIdx.x < 4) {
      double t = 0.0;
      for (int w = 0; w < OBS_ROWS; ++w) t += dots[w][threadIdx.x];
      coherent_store(&rec[threadIdx.x], t);
    } else if ((int)threadIdx.x < 4 + norb) {
      const int p = threadIdx.x - 4;
      double t = 0.0;
      for (int w = 0; w < OBS_ROWS; ++w) {
        const int64_t Aw = (int64_t)bx * OBS_ROWS + w;
        if (Aw < na && ((strs_a[Aw] >> p) & 1ull)) t += wrow[w];
      }
      coherent_store(&rec[4 + p], t);
    }
  } else {
    const int col = lane, rl = wv, RL = blockDim.x >> 6;
    const int64_t B = (int64_t)(bx - nrb) * 64 + col;
    double s = 0.0;
    if (B < nb)
      for (int64_t a0 = rl; a0 < na; a0 += (int64_t)RL * 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t a = a0 + (int64_t)u * RL;
          v[u] = C[(a < na ? a : a0) * nb + B];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += (a0 + (int64_t)u * RL < na) ? v[u] * v[u] : 0.0;
      }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0) {
      for (int r = 1; r < RL; ++r) s += red[r * 64 + col];
      wrow[col] = (B < nb) ? s : 0.0;
    }
    __syncthreads();
    if ((int)threadIdx.x < 4) {
      coherent_store(&rec[threadIdx.x], 0.0);
    } else if ((int)threadIdx.x < 4 + norb) {
      const int p = threadIdx.x - 4;
      double t = 0.0;
      for (int j = 0; j < 64; ++j) {
        const int64_t Bj = (int64_t)(bx - nrb) * 64 + j;
        if (Bj < nb && ((strs_b[Bj] >> p) & 1ull)) t += wrow[j];
      }
      coherent_store(&rec[4 + p], t);
    }
  }
  if (arrive_last(g.counter, bx, nbx)) observables_finish(g, nbx, red);
}

// the last workgroup's part of k_observables: fold the partial records, post the results and their sequence word
__device__ inline void observables_finish(const ObsArgs& g, unsigned nbx, double* red) {
  const int norb = g.norb;
  const unsigned nrb = g.nrb;
  double* partial = g.partial;
  double* out = g.out;
  // out = {c.Hc, c.S2c, c.c, occ_a[norb], occ_b[norb], |S2 c|^2}.  J threads share one result: thread (r, j) adds the
  // partial records b0 + j, b0 + j + J, ... (eight write-through loads in flight per round -- one thread per result
  // walked the records in ~2 us rounds, 10 us for 45 workgroups), the J sub-sums are added in order j = 0..J-1.
  const int nres = 3 + 2 * norb + 1;
  const int J = (int)blockDim.x / nres < 8 ? (int)blockDim.x / nres : 8;
  const int r = threadIdx.x / J, j = threadIdx.x % J;
  double t = 0.0;
  if (r < nres) {
    int field;
    unsigned b0, b1;
    if (r < 3) {
      field = r;
      b0 = 0;
      b1 = nrb;
    } else if (r < 3 + norb) {
      field = 4 + (r - 3);
      b0 = 0;
      b1 = nrb;
    } else if (r < 3 + 2 * norb) {
      field = 4 + (r - 3 - norb);
      b0 = nrb;
      b1 = nbx;
    } else {
      field = 3;
      b0 = 0;
      b1 = nrb;
    }
    for (unsigned b = b0 + j; b < b1; b += 8 * J) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = coherent_load(&partial[(int64_t)(b + u * J < b1 ? b + u * J : b) * OBS_W + field]);
#pragma unroll
      for (int u = 0; u < 8; ++u) t += (b + u * J < b1) ? v[u] : 0.0;
    }
  }
  red[threadIdx.x] = t;
  __syncthreads();
  if (r < nres && j == 0) {
    for (int k = 1; k < J; ++k) t += red[threadIdx.x + k];
    red[512 + r] = t;
  }
  __syncthreads();
  // results leave the chip from consecutive lanes: a few full-line writes over PCIe instead of one per result
  if ((int)threadIdx.x < nres) mail_store(&out[threadIdx.x], red[512 + threadIdx.x]);
  if (g.record) {
    if ((int)threadIdx.x < nres) g.record[1 + threadIdx.x] = red[512 + threadIdx.x];
    if (threadIdx.x == 0) {
      const DavState* st = g.st;
      g.record[0] = st ? st->e : 0.0;
    }
  }
  // sequence word behind the results: the host waits for it by reading memory instead of polling hipStreamQuery (whose
  // runtime lock other contexts' host threads need for their launches: 16 concurrent 317 x 317 solves ran 15-25 %
  // faster once the waiting threads stopped hammering it)
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    *static_cast<volatile long long*>(static_cast<long long*>(g.seq_word)) = g.seq;
  }
}
__global__ void k_observables(const ObsArgs g) { observables_body(g, blockIdx.x, gridDim.x); }
// batched (sqd_solve_batch): blockIdx.z = subspace, every subspace on the grid a single solve would give it
__global__ void k_observables_b(const ObsArgs* __restrict__ gs) {
  const ObsArgs g = gs[blockIdx.z];  // (a by-value copy: the record in SGPRs, as a kernel argument would be)
  if (blockIdx.x >= g.gx) return;
  observables_body(g, blockIdx.x, g.gx);
}

constexpr int OBS_MAIL = 3 * 128;  // doubles into the host-visible mailbox (slots 0..2 belong to the Davidson)
constexpr int OBS_SEQ = OBS_MAIL + 200;  // its sequence word (results: at most 3 + 2 * 64 + 1 doubles)
constexpr int OBS_STATE_SEQ = OBS_MAIL + 272;  // "the late copy of the state has landed" (its own 128-byte line)

int dev_observables(sqd_ctx* c, const double* d_c, double* out_host) {
  SQD_TRY(dev_observables_enqueue(c, d_c));
  SQD_TRY(dev_observables_wait(c));
  dev_observables_collect(c, out_host);
  return SQD_OK;
}
// wait for the latest k_observables: its sequence word first (a memory read per poll), then the stream itself, which is
// done or about to be (the kernel's other effects -- the state written to the caller's buffer -- count as complete
// only with the kernel)
// ---- the state to the caller's page-locked buffer, asynchronously (sqd_ctx_set_async_state): a kernel of its own on the
// context's COPY stream, started by an event behind the solution.  Posted full-line PCIe writes, 33 GB/s: 24 us for the
// 0.8 MB of a headline solve -- longer than any other kernel of that solve, and on the solver's stream it would sit in
// front of the next solve's table build.  The last workgroup posts the ticket behind everybody's stores.
__global__ void __launch_bounds__(256) k_state_copy(const double* __restrict__ C, double* __restrict__ host, int64_t n,
                                                     unsigned* counter, long long* state_word, long long seq) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
    double v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = C[i0 + u * stride < n ? i0 + u * stride : i0];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * stride < n) __builtin_nontemporal_store(v[u], &host[i0 + u * stride]);
  }
  // (one system-scope release per workgroup, behind everybody's stores: sqd_device.h, mailbox protocol)
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) __threadfence_system();
  if (!arrive_last(counter, blockIdx.x, gridDim.x)) return;
  if (threadIdx.x == 0) {
    __threadfence_system();
    *static_cast<volatile long long*>(state_word) = seq;
  }
}
// enqueue it for the resident solution; *ticket = what state_copy_wait takes
int state_copy_enqueue(sqd_ctx* c, double* host_twin, long long* ticket) {
  SQD_TRY(reserve_counters(c));
  c->state_seq = ++c->mail_seq;
  SQD_HIP_CHECK(hipEventRecord(c->ev_sol, c->stream));
  SQD_HIP_CHECK(hipStreamWaitEvent(c->copy_stream, c->ev_sol, 0));
  const int64_t n = c->D;
  unsigned blocks = (unsigned)((n + 1023) / 1024);
  if (blocks > 128) blocks = 128;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_state_copy, dim3(blocks), dim3(256), 0, c->copy_stream, c->sol.as<double>(), host_twin, n,
                     counter2_ptr(c), reinterpret_cast<long long*>(c->d_mail + OBS_STATE_SEQ), (long long)c->state_seq);
  SQD_HIP_CHECK(hipGetLastError());
  *ticket = (long long)c->state_seq;
  return SQD_OK;
}
bool state_copy_landed(const sqd_ctx* c, long long ticket) {
  return *reinterpret_cast<const volatile long long*>(c->h_mail + OBS_STATE_SEQ) >= ticket;
}
int dev_observables_wait(sqd_ctx* c, bool whole_kernel) {
  SQD_TRY(spin_wait_word(c->h_mail + OBS_SEQ, c->obs_seq, c->stream));
  return whole_kernel ? spin_stream_sync(c->stream) : SQD_OK;
}
// the late copy of the state with this ticket (or a later one: tickets grow with every solve of the context, and the
// stream runs them in order) has landed in the caller's buffer
int state_copy_wait(sqd_ctx* c, long long ticket) {
  return spin_wait_word(c->h_mail + OBS_STATE_SEQ, ticket, c->copy_stream);
}
// Called by every writer of `sol` other than sqd_solve's asynchronous path (which alternates sol / sol_alt and checks
// the other buffer's ticket itself): run_davidson reached through sqd_davidson or a solve whose state does not travel
// late, the row-sharded run, the timing hooks' memset.  If the latest asynchronous solve's k_state_copy is still
// reading `sol` on the copy stream, wait for it -- the earlier caller's SCIState would otherwise receive a mixture.
int sol_writer_guard(sqd_ctx* c) {
  if (c->sol_ticket > 0 && !state_copy_landed(c, c->sol_ticket)) SQD_TRY(state_copy_wait(c, c->sol_ticket));
  return SQD_OK;
}
void dev_observables_collect(sqd_ctx* c, double* out_host) {
  const int nres = 3 + 2 * c->norb + 1;
  for (int i = 0; i < nres; ++i) out_host[i] = c->h_mail[OBS_MAIL + i];
}
// arguments of k_observables for the state d_c of this subspace; t1 / t2: H c and S^2 c already built, or nullptr
static int fill_obs_args(sqd_ctx* c, const double* d_c, const double* t1, const double* t2, bool s2_inline,
                         double* host_twin, ObsArgs* gp, bool late_state = false) {
  ObsArgs& g = *gp;
  std::memset(&g.dg, 0, sizeof(g.dg));
  if (s2_inline) {
    fill_direct_args(c, d_c, nullptr, /*mode=*/1, /*spin=*/false, 0.0, 0.0, 0, 0, &g.dg);
    g.dg.stop = nullptr;
    g.dg.vec_index = nullptr;
  }
  const unsigned nrb = (unsigned)((c->na + OBS_ROWS - 1) / OBS_ROWS), ncb = (unsigned)((c->nb + 63) / 64);
  SQD_TRY(c->scratch.reserve((size_t)(nrb + ncb) * OBS_W * 8 + 64));
  SQD_TRY(reserve_counters(c));
  c->obs_seq = ++c->mail_seq;
  g.C = d_c;
  g.T1 = t1;
  g.T2 = t2;
  g.na = c->na;
  g.nb = c->nb;
  g.strs_a = c->sp[0].strs.as<uint64_t>();
  g.strs_b = c->sp[1].strs.as<uint64_t>();
  g.norb = c->norb;
  g.nrb = nrb;
  g.gx = nrb + ncb;
  g.partial = c->scratch.as<double>();
  g.counter = counter_ptr(c);
  g.out = c->d_mail + OBS_MAIL;
  g.host_c = late_state ? nullptr : host_twin;
  g.seq_word = reinterpret_cast<long long*>(c->d_mail + OBS_SEQ);
  g.seq = (long long)c->obs_seq;
  g.s2_inline = s2_inline ? 1 : 0;
  g.record = c->record_out;
  g.st = (c->record_out && c->have_solution) ? static_cast<const DavState*>(dav_state_ptr(c)) : nullptr;
  return SQD_OK;
}
static bool obs_s2_inline(const sqd_ctx* c, bool with_s2) {
  return with_s2 && c->sig_direct && c->sig_rows == 0 && !c->sig_lists && !c->sharded();
}
int dev_observables_enqueue(sqd_ctx* c, const double* d_c, bool with_h, bool with_s2, double* host_twin, bool late_state) {
  hipStream_t st = c->stream;
  const int64_t D = c->D;
  const double *t1 = nullptr, *t2 = nullptr;
  if (with_h) {
    SQD_TRY(c->tmp1.reserve((size_t)D * 8));
    SQD_TRY(launch_sigma(c, d_c, c->tmp1.as<double>(), 0, false, 0.0, 0.0));
    t1 = c->tmp1.as<double>();
  }
  const bool s2_inline = obs_s2_inline(c, with_s2);
  if (with_s2 && !s2_inline) {
    SQD_TRY(c->tmp2.reserve((size_t)D * 8));
    SQD_TRY(launch_sigma(c, d_c, c->tmp2.as<double>(), 1, false, 0.0, 0.0));
    t2 = c->tmp2.as<double>();
  }
  ObsArgs g;
  SQD_TRY(fill_obs_args(c, d_c, t1, t2, s2_inline, host_twin, &g, late_state));
  hipLaunchKernelGGL(k_observables, dim3(g.gx), dim3(512), 0, st, g);
  SQD_HIP_CHECK(hipGetLastError());
  return SQD_OK;
}

// ---- batched (sqd_solve_batch): the observables of every subspace's resident solution in one launch; with_s2 and a
// subspace outside the element-gather class: its S^2 c comes from a batched sigma launch in front (mode 1)
size_t observables_batch_bytes(size_t nsub) { return nsub * (sizeof(ObsArgs) + 64) + sigma_batch_bytes(nsub) + 256; }
int observables_batch_prepare(sqd_ctx* parent, const std::vector<sqd_ctx*>& subs, bool with_s2,
                              const std::vector<double*>& host_twin, char* h, char* d, size_t* off_io,
                              ObsBatchPlan* plan) {
  const int n = (int)subs.size();
  size_t off = (*off_io + 63) & ~size_t(63);
  ObsArgs* ha = reinterpret_cast<ObsArgs*>(h + off);
  plan->args = d + off;
  plan->n = n;
  plan->gx = 1;
  plan->have_s2_sigma = false;
  off += (size_t)n * sizeof(ObsArgs);
  std::vector<sqd_ctx*> ssubs;
  std::vector<const double*> sin;
  std::vector<double*> sout;
  for (int p = 0; p < n; ++p) {
    sqd_ctx* c = subs[p];
    const double* d_c = c->sol.as<double>();
    const bool inl = obs_s2_inline(c, with_s2);
    const double* t2 = nullptr;
    if (with_s2 && !inl) {
      SQD_TRY(c->tmp2.reserve((size_t)c->D * 8));
      t2 = c->tmp2.as<double>();
      ssubs.push_back(c);
      sin.push_back(d_c);
      sout.push_back(c->tmp2.as<double>());
    }
    SQD_TRY(fill_obs_args(c, d_c, nullptr, t2, inl, host_twin[p], &ha[p]));
    plan->gx = ha[p].gx > plan->gx ? ha[p].gx : plan->gx;
  }
  if (!ssubs.empty()) {
    SQD_TRY(sigma_batch_plan(ssubs, sin, sout, 1, false, 0.0, 0.0, 0, h, d, &off, &plan->s2_sigma));
    plan->have_s2_sigma = true;
  }
  *off_io = off;
  return SQD_OK;
}
int observables_batch_launch(sqd_ctx* parent, const ObsBatchPlan& plan) {
  if (plan.have_s2_sigma) SQD_TRY(sigma_batch_launch(parent, plan.s2_sigma));
  hipLaunchKernelGGL(k_observables_b, dim3(plan.gx, 1, (unsigned)plan.n), dim3(512), 0, parent->stream,
                     reinterpret_cast<const ObsArgs*>(plan.args));
  SQD_HIP_CHECK(hipGetLastError());
  return SQD_OK;
}

// dm2 pieces on the device.  resolved == false: spin-summed dm2 (pyscf make_rdm2) into out[0].
// resolved == true: (dm2aa, dm2ab, dm2bb) (pyscf make_rdm2s; dm2ab[p,q,r,s] = <p+_a r+_b s_b q_a>) into out[0..2].
static int rdm2_impl(sqd_ctx* c, const double* d_c, bool resolved, double* const* out) {
  const int norb = c->norb;
  const int64_t n2 = (int64_t)norb * norb, n4 = n2 * n2;
  hipStream_t st = c->stream;
  // layout of scratch: same[2][n4] (the second only when resolved) | G[n4] | per-spin { w[n] | dots_s[n_s] |
  // dots_d[n_d] } | xlinks
  const int64_t nhead = resolved ? 3 * n4 : 2 * n4;
  int64_t need = nhead;
  int64_t xl[2];
  for (int s = 0; s < 2; ++s) {
    need += c->sp[s].n + c->sp[s].n_s + c->sp[s].n_d;
    xl[s] = c->sp[s].n_s + c->sp[s].n * c->sp[s].nocc;
  }
  const size_t xbytes = (size_t)(xl[0] + xl[1]) * sizeof(XLink);
  // (+ the row form of the opposite-spin block: the lists once more, sorted; 2 x 2 x 4096 counters; its work items)
  const size_t max_items = 4096 + (size_t)xl[0] / RAB_E + 1;
  const size_t rbytes = xbytes + 4 * 4096 * sizeof(unsigned) + max_items * sizeof(RabItem) + 256;
  SQD_TRY(c->scratch.reserve((size_t)need * 8 + xbytes + rbytes + 256));
  double* base = c->scratch.as<double>();
  double* same[2] = {base, resolved ? base + n4 : base};
  double* G = base + nhead - n4;
  SQD_HIP_CHECK(hipMemsetAsync(base, 0, nhead * 8, st));
  double* p = base + nhead;
  XLink* xlink[2];
  xlink[0] = reinterpret_cast<XLink*>(base + need);
  xlink[1] = xlink[0] + xl[0];
  for (int s = 0; s < 2; ++s) {
    const SpinTables& t = c->sp[s];
    double* w = p;
    double* ds = p + t.n;
    double* dd = ds + t.n_s;
    p += t.n + t.n_s + t.n_d;
    SQD_TRY(spin_link_dots(c, s, d_c, w, ds, dd));
    hipLaunchKernelGGL(k_rdm2_diag, dim3(nblk(n2, 64)), dim3(64), 0, st, (const uint64_t*)t.strs.as<uint64_t>(), t.n,
                       (const double*)w, norb, same[s]);
    if (t.n_s > 0)
      hipLaunchKernelGGL(k_rdm2_singles, dim3(nblk(t.n_s, 256)), dim3(256), 0, st,
                         (const uint64_t*)t.strs.as<uint64_t>(), t.n_s, (const SRec*)t.s_rec.as<SRec>(),
                         (const double*)ds, norb, same[s]);
    if (t.n_d > 0)
      hipLaunchKernelGGL(k_rdm2_doubles, dim3(nblk(t.n_d, 256)), dim3(256), 0, st, t.n_d,
                         (const uint32_t*)t.d_orb.as<uint32_t>(), (const double*)dd, norb, same[s]);
    hipLaunchKernelGGL(k_xlinks, dim3(nblk(xl[s], 256)), dim3(256), 0, st, (const uint64_t*)t.strs.as<uint64_t>(), t.n,
                       t.nocc, t.n_s, (const uint32_t*)t.s_row.as<uint32_t>(), (const SRec*)t.s_rec.as<SRec>(),
                       xlink[s]);
    SQD_HIP_CHECK(hipGetLastError());
  }
  if (xl[0] > 0 && xl[1] > 0) {
    // connected sets: sorted lists + staged rows (k_rdm2_ab_rows); small or very long-rowed ones: a thread per beta xlink
    bool rows = (double)xl[0] * (double)xl[1] >= 2e8 && c->nb <= 65535 && c->na <= 0x7fffffff;
    int T = 1024;
    if (const char* env = std::getenv("SQD_RDM2_ROWS")) {  // test hook: "T" forces the row form with T threads, "0" forbids
      const int v = std::atoi(env);
      rows = v >= 64 && v <= 1024 && v % 64 == 0 && c->nb <= 65535;
      if (rows) T = v;
    }
    const size_t shmem = (size_t)(((c->nb + 1) & ~int64_t(1)) * 2) * 8;
    const int64_t H = (xl[1] + (int64_t)RAB_S * T - 1) / ((int64_t)RAB_S * T);
    if (shmem + 1024 > (size_t)c->lds_bytes || H > 4096) rows = false;
    if (rows) {
      // sort both lists by orbital pair (order inside a pair: as the atomics fall -- the sums below are atomic anyway)
      XLink* xs[2] = {xlink[1] + xl[1], xlink[1] + xl[1] + xl[0]};
      unsigned* hist = reinterpret_cast<unsigned*>(xs[1] + xl[1]);  // [2][4096] counts, then [2][4096] cursors
      unsigned* cursor = hist + 2 * 4096;
      RabItem* d_items = reinterpret_cast<RabItem*>(cursor + 2 * 4096);
      SQD_HIP_CHECK(hipMemsetAsync(hist, 0, 2 * 4096 * sizeof(unsigned), st));
      for (int sd = 0; sd < 2; ++sd) {
        hipLaunchKernelGGL(k_xl_hist, dim3(nblk(xl[sd], 256)), dim3(256), 0, st, (const XLink*)xlink[sd], xl[sd], hist + sd * 4096);
        hipLaunchKernelGGL(k_xl_scan, dim3(1), dim3(1024), 0, st, (const unsigned*)(hist + sd * 4096), cursor + sd * 4096);
        hipLaunchKernelGGL(k_xl_scatter, dim3(nblk(xl[sd], 256)), dim3(256), 0, st, (const XLink*)xlink[sd], xl[sd], cursor + sd * 4096, xs[sd]);
      }
      SQD_HIP_CHECK(hipGetLastError());
      std::vector<unsigned> h_hist(4096);
      SQD_HIP_CHECK(hipMemcpyAsync(h_hist.data(), hist, 4096 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
      SQD_STREAM_SYNC(st);
      std::vector<RabItem> items;
      uint32_t at = 0;
      for (int b = 0; b < 4096; ++b) {
        for (uint32_t o = 0; o < h_hist[b]; o += RAB_E) items.push_back(RabItem{at + o, std::min<uint32_t>(RAB_E, h_hist[b] - o)});
        at += h_hist[b];
      }
      if (items.size() > max_items) {
        set_error("internal: more opposite-spin rdm2 work items than planned");
        return SQD_ERR_STATE;
      }
      SQD_HIP_CHECK(hipMemcpyAsync(d_items, items.data(), items.size() * sizeof(RabItem), hipMemcpyHostToDevice, st));
      RabArgs ra;
      ra.c = d_c;
      ra.nb = c->nb;
      ra.lb = xl[1];
      ra.xa = xs[0];
      ra.xb = xs[1];
      ra.items = d_items;
      ra.n_items = (unsigned)items.size();
      ra.H = (int)H;
      ra.T = T;
      ra.norb = norb;
      ra.G = G;
      if (shmem > 64 * 1024)
        SQD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rdm2_ab_rows), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          (int)shmem));
      hipLaunchKernelGGL(k_rdm2_ab_rows, dim3((unsigned)(items.size() * (size_t)H)), dim3((unsigned)T), shmem, st, ra);
      SQD_HIP_CHECK(hipGetLastError());
      SQD_STREAM_SYNC(st);  // (the item list is a host vector copied asynchronously)
    } else {
      unsigned gy = (unsigned)(xl[0] < 1024 ? xl[0] : 1024);
      hipLaunchKernelGGL(k_rdm2_ab, dim3(nblk(xl[1], 256), gy), dim3(256), 0, st, d_c, c->nb, xl[0], xl[1],
                         (const XLink*)xlink[0], (const XLink*)xlink[1], norb, G);
    }
  }
  if (!resolved) {
    hipLaunchKernelGGL(k_rdm2_symm_add, dim3(nblk(n4, 256)), dim3(256), 0, st, norb, (const double*)G, same[0]);
    SQD_HIP_CHECK(hipGetLastError());
    SQD_HIP_CHECK(hipMemcpyAsync(out[0], same[0], n4 * 8, hipMemcpyDeviceToHost, st));
  } else {
    SQD_HIP_CHECK(hipGetLastError());
    SQD_HIP_CHECK(hipMemcpyAsync(out[0], same[0], n4 * 8, hipMemcpyDeviceToHost, st));
    SQD_HIP_CHECK(hipMemcpyAsync(out[1], G, n4 * 8, hipMemcpyDeviceToHost, st));
    SQD_HIP_CHECK(hipMemcpyAsync(out[2], same[1], n4 * 8, hipMemcpyDeviceToHost, st));
  }
  SQD_STREAM_SYNC(st);
  return SQD_OK;
}

int dev_rdm2(sqd_ctx* c, const double* d_c, double* dm2_host) {
  double* out[1] = {dm2_host};
  return rdm2_impl(c, d_c, false, out);
}
int dev_rdm2s(sqd_ctx* c, const double* d_c, double* dm2aa, double* dm2ab, double* dm2bb) {
  double* out[3] = {dm2aa, dm2ab, dm2bb};
  return rdm2_impl(c, d_c, true, out);
}

}  // namespace sqd
