// Same-spin part of sigma for CONNECTED string sets of 10^3 strings per spin and more (D = 10^6 .. 10^7):
//     G = H_a C + C H_b
// with H_a (na x na), H_b (nb x nb) the same-spin blocks of the projected Hamiltonian -- the Slater-Condon singles and
// doubles that pyscf's selected_ci.contract_2e evaluates through SCIcontract_2e_aaaa on C and on C^T (reference call
// sites qiskit_addon_sqd/fermion.py:721-723, :810-818).
//
// Why a third formulation.  Hartree-Fock-centred sets thin out as they grow: the same-spin blocks are 26 % dense at 317
// strings, 11 % at 1000, 5.6 % at 3000 (100 .. 160 links per string).  The matrix-core product (k_same_spin_mfma) pays
// 1 / density in flops at the SAME peak as the vector units (f64: 78.6 TFLOP/s either way on gfx950), and the work-item
// kernel evaluates the beta side as LDS gathers with one partial sum per 8 links.  Here both sides are row AXPYs with
// wave-uniform (scalar) coefficients and unit-stride operand rows, i.e. a sparse-matrix x dense-matrix product in its
// natural orientation:
//     G[A, :]   = sum_l  val[l] * C  [src[l], :]      alpha: on C itself
//     G2T[B, :] = sum_l  val[l] * C^T[src[l], :]      beta : the same kernel on the transposed vector
// between two tiled transpositions (C -> C^T, and G += G2T^T), 16 + 24 bytes per element, cache resident at these sizes.
// No gathers, no LDS, no partial rows; the link records are read through the scalar cache; the operand rows come out of
// the L2 of the XCD that owns the column panel (panel p is processed on XCD p mod 8 only, so an XCD's L2 holds n x 64 J
// doubles of one panel at a time).  Bound: bytes through the vector L1 (64 B / clk / CU: 8 multiply-adds per clock and
// CU).  The work items of k_sigma then add G element by element exactly as they add the matrix-core product (dense
// same-spin mode with ONE partial product).  Fixed order of accumulation: the same bits on every run.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <numeric>

#include "sqd_common.h"

namespace sqd {

constexpr int SPMM_U = 16;  // the merged lists are padded to multiples of 16 links with zero-weight links (rounds of 8 or 16)
// element i (a 32-bit lane offset) of a row whose address is wave-uniform: scalar base + 32-bit byte offset, no 64-bit
// address arithmetic per lane and load
__device__ inline double spmm_ldu(const double* base, unsigned i) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + (i << 3));
}

struct SpmmSide {
  DevBuf ptr, src, val, order;
  std::vector<uint32_t> h_order;  // host copies: the uploads are asynchronous
  std::vector<int64_t> h_ptr;     // padded row pointers
  int64_t n = 0, m = 0, links = 0;
};
// row-grouped form (k_spmm_grouped): GR adjacent rows share ONE sorted list of sources (the union of their lists) with a
// dense GR-vector of coefficients per source
struct SpmmGroups {
  DevBuf base, cnt, src, coef, order;
  std::vector<int64_t> h_base;
  std::vector<uint32_t> h_order;
  int64_t ngroups = 0, cap = 0;
};
struct SpmmState {
  SpmmSide side[2];
  SpmmGroups groups[2];
  bool grouped = false;
  DevBuf ct, g2t;  // C^T (nb x na) and H_b C^T (nb x na)
};

#ifndef SQD_SPMM_GR
#define SQD_SPMM_GR 8  // (probe builds: 4 / 16 -- profiles/r05/spmm_group_size_probe.txt)
#endif
int spmm_rows_per_group(const sqd_ctx* c) {
  return c->spmm && static_cast<const SpmmState*>(c->spmm)->grouped ? SQD_SPMM_GR : 1;
}
void spmm_release(sqd_ctx* c) {
  if (!c->spmm) return;
  SpmmState* s = static_cast<SpmmState*>(c->spmm);
  for (auto& sd : s->side)
    for (DevBuf* b : {&sd.ptr, &sd.src, &sd.val, &sd.order}) b->release();
  for (auto& gr : s->groups)
    for (DevBuf* b : {&gr.base, &gr.cnt, &gr.src, &gr.coef, &gr.order}) b->release();
  s->ct.release();
  s->g2t.release();
  delete s;
  c->spmm = nullptr;
}

// ---- merged same-spin CSR of one spin: row i = its single links (value incl. sign and the mean-field part), then its
// double links, then zero-weight padding (source = the row itself) up to a whole round of SPMM_U links; one wavefront
// per row (rows of the Hartree-Fock neighbourhood hold over a thousand links).  The padded row pointers are cut on the
// host from the CSR pointers it holds anyway.
struct MergeArgs {
  int64_t n[2];
  GPtr<const int64_t> s_ptr[2], d_ptr[2];
  GPtr<const SRec> s_rec[2];
  GPtr<const double> s_val[2], d_val[2];
  GPtr<const uint32_t> d_src[2];
  GPtr<const int64_t> ptr[2];
  GPtr<uint32_t> src[2];
  GPtr<double> val[2];
};
__global__ void __launch_bounds__(256) k_spmm_merge(const MergeArgs g) {
  const int s = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i > g.n[s]) return;
  uint32_t* __restrict__ src = g.src[s];
  double* __restrict__ val = g.val[s];
  const int64_t o = g.ptr[s][i];
  if (i == g.n[s]) {  // the round of padding behind the last row (the kernel's record prefetch reads it)
    if (lane < SPMM_U) {
      src[o + lane] = 0u;
      val[o + lane] = 0.0;
    }
    return;
  }
  const int64_t* __restrict__ sp = g.s_ptr[s];
  const int64_t* __restrict__ dp = g.d_ptr[s];
  const int64_t s0 = sp[i], ns = sp[i + 1] - s0, d0 = dp[i], nd = dp[i + 1] - d0;
  const int64_t o1 = g.ptr[s][i + 1];
  const SRec* __restrict__ rec = g.s_rec[s];
  const double* __restrict__ sv = g.s_val[s];
  const uint32_t* __restrict__ ds = g.d_src[s];
  const double* __restrict__ dv = g.d_val[s];
  for (int64_t k = lane; k < ns; k += 64) {
    src[o + k] = rec[s0 + k].src;
    val[o + k] = sv[s0 + k];
  }
  for (int64_t k = lane; k < nd; k += 64) {
    src[o + ns + k] = ds[d0 + k];
    val[o + ns + k] = dv[d0 + k];
  }
  for (int64_t k = o + ns + nd + lane; k < o1; k += 64) {
    src[k] = (uint32_t)i;
    val[k] = 0.0;
  }
}

// ---- tiled transposition, 64 x 64 doubles through LDS (pitch 65: conflict-free both ways).  add = 0: out[c][r] =
// in[r][c]; add = 1: out[c][r] += in[r][c].  The input of the first transposition is the vector the Davidson run
// selected on the device (vec_index), like every sigma kernel's.
struct TransArgs {
  GPtr<const double> in;
  GPtr<double> out;
  int64_t rows, cols;  // of `in`
  int add;
  GPtr<const int> stop, vec_index;
  int64_t in_stride;
};
__global__ void __launch_bounds__(256) k_spmm_transpose(const TransArgs g) {
  __shared__ double tile[64 * 65];
  if (g.stop && *g.stop) return;
  const double* __restrict__ in = g.in + (g.vec_index ? (int64_t)(*g.vec_index - 1) * g.in_stride : 0);
  double* __restrict__ out = g.out;
  const int64_t r0 = (int64_t)blockIdx.y * 64, c0 = (int64_t)blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  double v[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t r = r0 + ty + 4 * k, cc = c0 + tx;
    v[k] = (r < g.rows && cc < g.cols) ? in[r * g.cols + cc] : 0.0;
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) tile[(ty + 4 * k) * 65 + tx] = v[k];
  __syncthreads();
  if (g.add) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int64_t orow = c0 + ty + 4 * k, ocol = r0 + tx;
      v[k] = (orow < g.cols && ocol < g.rows) ? out[orow * g.rows + ocol] : 0.0;
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int64_t orow = c0 + ty + 4 * k, ocol = r0 + tx;
    if (orow < g.cols && ocol < g.rows) {
      const double t = tile[tx * 65 + ty + 4 * k];
      out[orow * g.rows + ocol] = g.add ? v[k] + t : t;
    }
  }
}

// ---- the product.  One wavefront per task (target row, panel of 64 J columns): the row's links in rounds of eight -- eight
// wave-uniform records, then the 8 J operand loads (512 consecutive bytes each) in flight together, then the multiply-adds.
// The padding of the last round re-reads the row's last link with weight zero (unconditional loads: exact wait counts).
// Rows are taken in the order of descending list length, panel after panel.
struct SpmmArgs {
  GPtr<const int64_t> ptr[2];
  GPtr<const uint32_t> src[2];
  GPtr<const double> val[2];
  GPtr<const uint32_t> order[2];
  GPtr<const double> in[2];
  GPtr<double> out[2];
  int64_t n[2], m[2];  // side s: n[s] rows of m[s] columns
  unsigned npanels[2];
  int xcd_split;  // 1: panel p is processed by the workgroups of XCD p mod 8 (workgroup b runs on XCD b mod 8)
  GPtr<const int> stop, vec_index;
  int64_t in_stride;  // side 0's input: the vector selected on the device
};
template <int J, int U>
__global__ void __launch_bounds__(256) k_spmm_rows(const SpmmArgs g) {
  if (g.stop && *g.stop) return;
  const int side = blockIdx.y;
  const uint32_t n = (uint32_t)g.n[side];
  const int64_t m = g.m[side];
  const unsigned np = g.npanels[side];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned panel, r;
  if (g.xcd_split) {
    const unsigned x = blockIdx.x & 7u, q = (blockIdx.x >> 3) * 4u + (unsigned)wave;
    panel = (q / n) * 8u + x;
    r = q % n;
  } else {
    const unsigned q = blockIdx.x * 4u + (unsigned)wave;
    panel = q / n;
    r = q % n;
  }
  if (panel >= np) return;  // (uniform over the wavefront)
  // (wave-uniform by construction; said so to the compiler, which sees threadIdx in them: scalar registers, scalar
  // loads of the link records, scalar row bases)
  panel = (unsigned)__builtin_amdgcn_readfirstlane((int)panel);
  r = (unsigned)__builtin_amdgcn_readfirstlane((int)r);
  const int64_t t = (int64_t)__builtin_amdgcn_readfirstlane((int)g.order[side][r]);
  const int64_t* __restrict__ ptr = g.ptr[side];
  const uint32_t* __restrict__ src = g.src[side];
  const double* __restrict__ val = g.val[side];
  const double* __restrict__ in = g.in[side];
  if (side == 0 && g.vec_index) in += (int64_t)(*g.vec_index - 1) * g.in_stride;
  const int64_t l0 = ptr[t], l1 = ptr[t + 1];  // (multiples of U: rows are padded with zero-weight links)
  const unsigned c0 = panel * (unsigned)(64 * J);
  unsigned col[J];  // column within the panel's segment, clamped (dead lanes shadow the last column and store nothing)
  bool ok[J];
  double acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int64_t cc = (int64_t)c0 + j * 64 + lane;
    ok[j] = cc < m;
    col[j] = (unsigned)((ok[j] ? cc : m - 1) - c0);
    acc[j] = 0.0;
  }
  in += c0;
  // records of the NEXT round are requested (scalar loads) before this round's operand rows: their round trip hides
  // behind the vector loads.  (The arrays end with one round of padding, so the last prefetch stays inside them.)
  uint32_t sn[U];
  double wn[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    sn[u] = src[l0 + u];
    wn[u] = val[l0 + u];
  }
  for (int64_t l = l0; l < l1; l += U) {
    uint32_t s[U];
    double w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      s[u] = sn[u];
      w[u] = wn[u];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      sn[u] = src[l + U + u];
      wn[u] = val[l + U + u];
    }
    double x[U][J];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const double* __restrict__ rowp = in + (int64_t)s[u] * m;  // (uniform: a scalar base)
#pragma unroll
      for (int j = 0; j < J; ++j) x[u][j] = spmm_ldu(rowp, col[j]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < J; ++j) acc[j] += w[u] * x[u][j];
  }
  double* __restrict__ out = g.out[side] + t * m + c0;
#pragma unroll
  for (int j = 0; j < J; ++j)
    if (ok[j]) out[col[j]] = acc[j];
}

// ---- the product on ROW GROUPS (the default).  Strings that are neighbours in the sorted order differ in their low
// orbitals only and are linked to nearly the same strings: the union of the source lists of GR = 8 adjacent rows is
// 3.4 (1000 strings) to 4.1 (3000) times one list, not 8 times.  A group therefore walks ONE list -- the union, sorted --
// with a dense 8-vector of coefficients per source (zeros where a row has no such link): an operand row segment that
// k_spmm_rows reads once per link is read once per group and serves 8 multiply-adds, 2.4 / 1.9 of them useful.  The
// bytes through the vector L1 per useful multiply-add drop to 0.42 / 0.52 of k_spmm_rows'; the extra multiply-adds by
// zero are free (the vector units have an 8-fold margin over the L1 here).  Records travel through the scalar cache:
// per round of GU sources one s_load of their addresses and GU x 8 coefficients, which enter v_fmac_f64 as scalar operands.
constexpr int GR = SQD_SPMM_GR, GPAD = 16;
struct GroupBuildArgs {
  int64_t n[2];
  GPtr<const int64_t> s_ptr[2], d_ptr[2];
  GPtr<const SRec> s_rec[2];
  GPtr<const double> s_val[2], d_val[2];
  GPtr<const uint32_t> d_src[2];
  GPtr<const int64_t> base[2];  // first source slot of every group (host: an upper bound of the union sizes)
  GPtr<uint32_t> cnt[2], src[2];
  GPtr<double> coef[2];         // [slot][GR], zeroed beforehand
};
// one wavefront per group: bitmap of the sources of its rows (LDS), ranks from the per-word popcount prefix, then the
// union's addresses and every link's value to coef[rank][row]
__global__ void __launch_bounds__(256) k_spmm_group_build(const GroupBuildArgs g) {
  __shared__ uint32_t bits[4][1024];   // up to 32 768 strings
  __shared__ uint32_t rank0[4][1024];  // sources below the word
  const int sd = blockIdx.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n = g.n[sd];
  const int64_t grp = (int64_t)blockIdx.x * 4 + w;
  const int64_t ngroups = (n + GR - 1) / GR;
  const bool live = grp < ngroups;
  const int nwords = (int)((n + 31) / 32);
  for (int i = lane; i < 1024; i += 64) bits[w][i] = 0u;
  __syncthreads();
  const int64_t* __restrict__ sp = g.s_ptr[sd];
  const int64_t* __restrict__ dp = g.d_ptr[sd];
  const SRec* __restrict__ rec = g.s_rec[sd];
  const uint32_t* __restrict__ ds = g.d_src[sd];
  const int64_t r0 = grp * GR, r1 = live ? (r0 + GR < n ? r0 + GR : n) : r0;
  if (live) {
    // (the rows of a group are consecutive: their single links, and their double links, are two contiguous ranges)
    for (int64_t k = sp[r0] + lane; k < sp[r1]; k += 64) atomicOr(&bits[w][rec[k].src >> 5], 1u << (rec[k].src & 31u));
    for (int64_t k = dp[r0] + lane; k < dp[r1]; k += 64) atomicOr(&bits[w][ds[k] >> 5], 1u << (ds[k] & 31u));
  }
  __syncthreads();
  // lane l owns words 16 l .. 16 l + 15
  uint32_t mine = 0;
  for (int u = 0; u < 16; ++u) mine += __popc(bits[w][16 * lane + u]);
  uint32_t incl = mine;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl(incl, lane - d);
    if (lane >= d) incl += y;
  }
  const uint32_t total = __shfl(incl, 63);
  uint32_t run = incl - mine;
  for (int u = 0; u < 16; ++u) {
    rank0[w][16 * lane + u] = run;
    run += __popc(bits[w][16 * lane + u]);
  }
  __syncthreads();
  if (!live) return;
  const int64_t base = g.base[sd][grp];
  const uint32_t padded = (total + GPAD - 1) / GPAD * GPAD;
  if (lane == 0) g.cnt[sd][grp] = padded;
  uint32_t* __restrict__ src = g.src[sd] + base;
  double* __restrict__ coef = g.coef[sd] + base * GR;
  for (int i = lane; i < nwords; i += 64) {
    uint32_t b = bits[w][i], r = rank0[w][i];
    while (b) {
      const int bit = __ffs((int)b) - 1;
      src[r++] = (uint32_t)(i * 32 + bit);
      b &= b - 1u;
    }
  }
  for (uint32_t p = total + lane; p < padded; p += 64) src[p] = 0u;  // (padding: any valid row, all-zero coefficients)
  const double* __restrict__ sv = g.s_val[sd];
  const double* __restrict__ dv = g.d_val[sd];
  for (int64_t r = r0; r < r1; ++r) {
    for (int64_t k = sp[r] + lane; k < sp[r + 1]; k += 64) {
      const uint32_t a = rec[k].src;
      const uint32_t rk = rank0[w][a >> 5] + __popc(bits[w][a >> 5] & ((1u << (a & 31u)) - 1u));
      coef[(int64_t)rk * GR + (r - r0)] = sv[k];
    }
    for (int64_t k = dp[r] + lane; k < dp[r + 1]; k += 64) {
      const uint32_t a = ds[k];
      const uint32_t rk = rank0[w][a >> 5] + __popc(bits[w][a >> 5] & ((1u << (a & 31u)) - 1u));
      coef[(int64_t)rk * GR + (r - r0)] = dv[k];
    }
  }
}

struct GroupedArgs {
  GPtr<const int64_t> base[2];
  GPtr<const uint32_t> cnt[2], src[2], order[2];
  GPtr<const double> coef[2];
  GPtr<const double> in[2];
  GPtr<double> out[2];
  int64_t n[2], m[2];
  unsigned ngroups[2], npanels[2];
  int xcd_split;
  GPtr<const int> stop, vec_index;
  int64_t in_stride;
};
template <int GX, int GU, int GJ>
__global__ void __launch_bounds__(256) k_spmm_grouped(const GroupedArgs g) {
  if (g.stop && *g.stop) return;
  const int side = blockIdx.y;
  const unsigned ng = g.ngroups[side], np = g.npanels[side];
  const int64_t n = g.n[side], m = g.m[side];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  unsigned panel, r;
  if (g.xcd_split) {
    const unsigned x = blockIdx.x & 7u, q = (blockIdx.x >> 3) * 4u + (unsigned)wave;
    panel = (q / ng) * 8u + x;
    r = q % ng;
  } else {
    const unsigned q = blockIdx.x * 4u + (unsigned)wave;
    panel = q / ng;
    r = q % ng;
  }
  if (panel >= np) return;  // (uniform over the wavefront)
  panel = (unsigned)__builtin_amdgcn_readfirstlane((int)panel);
  r = (unsigned)__builtin_amdgcn_readfirstlane((int)r);
  const int64_t grp = (int64_t)__builtin_amdgcn_readfirstlane((int)g.order[side][r]);
  const double* __restrict__ in = g.in[side];
  if (side == 0 && g.vec_index) in += (int64_t)(*g.vec_index - 1) * g.in_stride;
  const int64_t base = g.base[side][grp];
  const uint32_t cnt = g.cnt[side][grp];  // (a multiple of GPAD, itself a multiple of GX)
  const uint32_t* __restrict__ src = g.src[side] + base;
  const double* __restrict__ coef = g.coef[side] + base * GR;
  // GJ columns per lane (panel = 64 GJ columns): a coefficient record fetched through the scalar cache serves GJ times
  // the multiply-adds
  const unsigned c0 = panel * (unsigned)(64 * GJ);
  bool ok[GJ];
  unsigned col[GJ];
#pragma unroll
  for (int j = 0; j < GJ; ++j) {
    ok[j] = (int64_t)c0 + j * 64 + lane < m;
    col[j] = ok[j] ? (unsigned)(j * 64 + lane) : (unsigned)(m - 1 - c0);
  }
  in += c0;
  double acc[GR][GJ];
#pragma unroll
  for (int i = 0; i < GR; ++i)
#pragma unroll
    for (int j = 0; j < GJ; ++j) acc[i][j] = 0.0;
  // GX operand rows in flight per wavefront (their addresses from one scalar load a chunk ahead); the coefficients
  // stream behind them GU sources at a time
  uint32_t sn[GX];
#pragma unroll
  for (int u = 0; u < GX; ++u) sn[u] = src[u];
  for (uint32_t l = 0; l < cnt; l += GX, src += GX, coef += GX * GR) {
    uint32_t s[GX];
    double x[GX][GJ];
#pragma unroll
    for (int u = 0; u < GX; ++u) s[u] = sn[u];
#pragma unroll
    for (int u = 0; u < GX; ++u) sn[u] = src[GX + u];  // (the list ends with one chunk of padding)
#pragma unroll
    for (int u = 0; u < GX; ++u)
#pragma unroll
      for (int j = 0; j < GJ; ++j) x[u][j] = spmm_ldu(in + (int64_t)s[u] * m, col[j]);
#pragma unroll
    for (int u0 = 0; u0 < GX; u0 += GU) {
      double cf[GU][GR];
#pragma unroll
      for (int u = 0; u < GU; ++u)
#pragma unroll
        for (int i = 0; i < GR; ++i) cf[u][i] = coef[(u0 + u) * GR + i];
#pragma unroll
      for (int u = 0; u < GU; ++u)
#pragma unroll
        for (int i = 0; i < GR; ++i)
#pragma unroll
          for (int j = 0; j < GJ; ++j) acc[i][j] += cf[u][i] * x[u0 + u][j];
    }
  }
  double* __restrict__ out = g.out[side] + c0;
#pragma unroll
  for (int j = 0; j < GJ; ++j)
    if (ok[j])
#pragma unroll
      for (int i = 0; i < GR; ++i)
        if (grp * GR + i < n) out[(grp * GR + i) * m + col[j]] = acc[i][j];
}

// ---- host side
// Is this subspace taken by the sparse-product same-spin path?  (phase 2 of set_subspace; single builds, whole row
// range.)  SQD_SIGMA_SPMM=1 / 0 forces / forbids (the tests run every same-spin formulation on the same inputs).
bool spmm_select(sqd_ctx* c, int64_t na, int64_t nb, int64_t row0, int64_t row1, const int64_t* tot, bool direct) {
  c->sig_spmm = false;
  if (direct || row0 != 0 || row1 != na) return false;
  if (na > 0xfffffffell / 64 || nb > 0xfffffffell / 64) return false;
  // (More than 8192 beta strings: the whole-row opposite-spin kernel does not take such rows; the work items run behind the
  // product -- through the multi-pass instantiation of k_sigma, see launch_sigma_r.)
  const char* env = std::getenv("SQD_SIGMA_SPMM");
  bool on = false;
  if (env) {
    on = std::atoi(env) != 0;
  } else {
    // connected sets from ~900 strings per spin.  Measured on the MI355X (profiles/r05/, HF-centred N x N, us per sigma,
    // this path with the whole-row opposite-spin kernel of sqd_opp.hip | matrix cores + work items | sparse work items):
    // 1000: 195 | 237 | 257; 2000: 763 | 1305 | 1244; 3000: 2209 | 3747 | 5073 (500 / 700 with the work items behind the
    // product: 98 | 67 | 73 and 145 | 111 | 125).  Smaller sets stay with the matrix cores: one launch instead of three,
    // and their blocks are 11-26 % dense.
    // Blocks more than a quarter full (near-complete string sets of few orbitals) go back to the matrix cores: all 1001
    // strings of (14o, 4e), 31 % dense, 911 us here against 699 us there (profiles/r05/dense_small_orbital_probe.txt).
    const int64_t same_a = tot[0] + tot[1], same_b = tot[2] + tot[3];
    // Unequal sides (profiles/r05/shape_probe*.txt, us per sigma, this path | matrix cores + work items | work items): a long
    // beta side carries it from ~300 alpha strings (300 x 3000: 325 | 324 | 535; 500 x 2000: 269 | 299 | 310; 600 x 6000:
    // 1396 | - | 3132), a long alpha side from ~450 beta strings (4000 x 500: 481 | 693 | 612; 2500 x 700: 412 | 471 | 445;
    // 6000 x 600: 1164 | - | 1377); short sides below that stay where they were (200 x 2000: 196 | 131 | 157).
    // ... for well-connected sets only (>= 4 single links per string on both sides: half-uniform sets of 300 x 1800 lose,
    // 134 us against 75)
    const bool unequal = ((nb >= 1800 && na >= 280) || (na >= 2500 && nb >= 450)) && tot[0] >= 4 * na && tot[2] >= 4 * nb;
    const bool sized = (na >= 896 && nb >= 896) || unequal;
    on = sized && same_a >= 8 * na && same_b >= 8 * nb && 4 * same_a <= na * na && 4 * same_b <= nb * nb;
  }
  c->sig_spmm = on;
  return on;
}

// columns per lane (panel width 64 J).  Measured on the MI355X (profiles/r05/spmm_sweep_probe.txt, HF-centred N x N, us
// per sigma with the work items idle): 1000: J = 1 166-173 | 2 185-202 | 4 203-291; 3000: 1674-1712 | 1717-2029 |
// 1844-2166 -- one column per lane: twice the wavefronts per row, each with 8 operand rows of 512 bytes in flight;
// the XCD split and the row order move nothing (the kernel is bound by the stream of operand rows out of the L2s,
// ~17 TB/s, whichever way the tasks are dealt)
static int spmm_J(int64_t, int64_t, bool) {
  if (const char* env = std::getenv("SQD_SPMM_J")) {  // tuning hook
    const int v = std::atoi(env);
    if (v == 1 || v == 2 || v == 4) return v;
  }
  return 1;
}

int spmm_build(sqd_ctx* c) {
  if (!c->spmm) c->spmm = new SpmmState();
  SpmmState* s = static_cast<SpmmState*>(c->spmm);
  const int64_t na = c->na, nb = c->nb;
  SQD_TRY(s->ct.reserve((size_t)na * nb * 8));
  SQD_TRY(s->g2t.reserve((size_t)na * nb * 8));
  SQD_TRY(c->gdense.reserve((size_t)na * nb * 8));
  MergeArgs ma;
  const int64_t* hptr[2][2] = {{c->h_sptr, c->h_dptr}, {c->h_sptr_b, c->h_dptr_b}};
  for (int sp = 0; sp < 2; ++sp) {
    const SpinTables& t = c->sp[sp];
    SpmmSide& d = s->side[sp];
    d.n = t.n;
    d.m = sp ? na : nb;
    const int64_t* ps = hptr[sp][0];
    const int64_t* pd = hptr[sp][1];
    d.h_ptr.resize((size_t)t.n + 1);
    d.h_ptr[0] = 0;
    for (int64_t i = 0; i < t.n; ++i) {
      const int64_t len = (ps[i + 1] - ps[i]) + (pd[i + 1] - pd[i]);
      d.h_ptr[i + 1] = d.h_ptr[i] + (len + SPMM_U - 1) / SPMM_U * SPMM_U;
    }
    d.links = d.h_ptr[t.n];
    SQD_TRY(d.ptr.reserve((size_t)(t.n + 1) * 8));
    SQD_TRY(d.src.reserve((size_t)(d.links + SPMM_U) * 4 + 64));
    SQD_TRY(d.val.reserve((size_t)(d.links + SPMM_U) * 8 + 64));
    SQD_HIP_CHECK(hipMemcpyAsync(d.ptr.p, d.h_ptr.data(), (size_t)(t.n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    // rows by descending list length (stable): the long rows of the Hartree-Fock neighbourhood start first
    d.h_order.resize((size_t)t.n);
    std::iota(d.h_order.begin(), d.h_order.end(), 0u);
    static const bool natural = [] {  // tuning hook: rows in string order (neighbours share sources: L1 hits)
      const char* env = std::getenv("SQD_SPMM_ORDER");
      return env && std::atoi(env) == 0;
    }();
    if (!natural) std::stable_sort(d.h_order.begin(), d.h_order.end(), [&](uint32_t a, uint32_t b) {
      return (ps[a + 1] - ps[a]) + (pd[a + 1] - pd[a]) > (ps[b + 1] - ps[b]) + (pd[b + 1] - pd[b]);
    });
    SQD_TRY(d.order.reserve((size_t)t.n * 4 + 16));
    SQD_HIP_CHECK(hipMemcpyAsync(d.order.p, d.h_order.data(), (size_t)t.n * 4, hipMemcpyHostToDevice, c->stream));
    ma.n[sp] = t.n;
    ma.s_ptr[sp] = t.s_ptr.as<int64_t>();
    ma.d_ptr[sp] = t.d_ptr.as<int64_t>();
    ma.s_rec[sp] = t.s_rec.as<SRec>();
    ma.s_val[sp] = t.s_val.as<double>();
    ma.d_src[sp] = t.d_src.as<uint32_t>();
    ma.d_val[sp] = t.d_val.as<double>();
    ma.ptr[sp] = d.ptr.as<int64_t>();
    ma.src[sp] = d.src.as<uint32_t>();
    ma.val[sp] = d.val.as<double>();
  }
  const int64_t maxn = na > nb ? na : nb;
  hipLaunchKernelGGL(k_spmm_merge, dim3((unsigned)((maxn + 1 + 3) / 4), 2), dim3(256), 0, c->stream, ma);
  SQD_HIP_CHECK(hipGetLastError());
  // the row-grouped form (default; SQD_SPMM_GROUPED=0 forbids): up to 32 768 strings per spin (the build kernel's bitmap)
  const char* genv = std::getenv("SQD_SPMM_GROUPED");  // (read per build: the tests switch it inside one process)
  const bool grouped_env = !genv || std::atoi(genv) != 0;
  s->grouped = grouped_env && maxn <= 32768;
  if (s->grouped) {
    GroupBuildArgs gb;
    for (int sp = 0; sp < 2; ++sp) {
      const SpinTables& t = c->sp[sp];
      SpmmGroups& gr = s->groups[sp];
      const int64_t* ps = hptr[sp][0];
      const int64_t* pd = hptr[sp][1];
      gr.ngroups = (t.n + GR - 1) / GR;
      gr.h_base.resize((size_t)gr.ngroups + 1);
      gr.h_order.resize((size_t)gr.ngroups);
      gr.h_base[0] = 0;
      for (int64_t q = 0; q < gr.ngroups; ++q) {
        const int64_t a = q * GR, b = (a + GR < t.n) ? a + GR : t.n;
        int64_t len = (ps[b] - ps[a]) + (pd[b] - pd[a]);  // the union holds at most this many sources, and at most n
        if (len > t.n) len = t.n;
        gr.h_base[q + 1] = gr.h_base[q] + (len + GPAD - 1) / GPAD * GPAD + GPAD;
      }
      gr.cap = gr.h_base[gr.ngroups];
      std::iota(gr.h_order.begin(), gr.h_order.end(), 0u);
      std::stable_sort(gr.h_order.begin(), gr.h_order.end(), [&](uint32_t a, uint32_t b) {
        return gr.h_base[a + 1] - gr.h_base[a] > gr.h_base[b + 1] - gr.h_base[b];
      });
      SQD_TRY(gr.base.reserve((size_t)(gr.ngroups + 1) * 8));
      SQD_TRY(gr.cnt.reserve((size_t)gr.ngroups * 4 + 16));
      SQD_TRY(gr.order.reserve((size_t)gr.ngroups * 4 + 16));
      SQD_TRY(gr.src.reserve((size_t)(gr.cap + GPAD) * 4 + 64));
      SQD_TRY(gr.coef.reserve((size_t)(gr.cap + GPAD) * GR * 8 + 64));
      SQD_HIP_CHECK(hipMemcpyAsync(gr.base.p, gr.h_base.data(), (size_t)(gr.ngroups + 1) * 8, hipMemcpyHostToDevice, c->stream));
      SQD_HIP_CHECK(hipMemcpyAsync(gr.order.p, gr.h_order.data(), (size_t)gr.ngroups * 4, hipMemcpyHostToDevice, c->stream));
      SQD_HIP_CHECK(hipMemsetAsync(gr.coef.p, 0, (size_t)(gr.cap + GPAD) * GR * 8, c->stream));
      gb.n[sp] = t.n;
      gb.s_ptr[sp] = t.s_ptr.as<int64_t>();
      gb.d_ptr[sp] = t.d_ptr.as<int64_t>();
      gb.s_rec[sp] = t.s_rec.as<SRec>();
      gb.s_val[sp] = t.s_val.as<double>();
      gb.d_src[sp] = t.d_src.as<uint32_t>();
      gb.d_val[sp] = t.d_val.as<double>();
      gb.base[sp] = gr.base.as<int64_t>();
      gb.cnt[sp] = gr.cnt.as<uint32_t>();
      gb.src[sp] = gr.src.as<uint32_t>();
      gb.coef[sp] = gr.coef.as<double>();
    }
    const int64_t maxg = (maxn + GR - 1) / GR;
    hipLaunchKernelGGL(k_spmm_group_build, dim3((unsigned)((maxg + 3) / 4), 2), dim3(256), 0, c->stream, gb);
    SQD_HIP_CHECK(hipGetLastError());
    return SQD_OK;
  }
  return SQD_OK;  // (more than 32 768 strings per spin, or SQD_SPMM_GROUPED=0: k_spmm_rows on the merged lists above)
}

// G (sqd_ctx::gdense, one partial product) = H_a C + C H_b for the vector the work items of the same sigma build read
int spmm_launch(sqd_ctx* c, const double* d_c, int64_t in_stride) {
  SpmmState* s = static_cast<SpmmState*>(c->spmm);
  if (!s) {
    set_error("internal: sparse-product same-spin mode without its tables");
    return SQD_ERR_STATE;
  }
  const int64_t na = c->na, nb = c->nb;
  const int* vidx = (c->sigma_index && in_stride) ? c->sigma_index : nullptr;
  TransArgs t1;
  t1.in = d_c;
  t1.out = s->ct.as<double>();
  t1.rows = na;
  t1.cols = nb;
  t1.add = 0;
  t1.stop = c->sigma_stop;
  t1.vec_index = vidx;
  t1.in_stride = in_stride;
  hipLaunchKernelGGL(k_spmm_transpose, dim3((unsigned)((nb + 63) / 64), (unsigned)((na + 63) / 64)), dim3(256), 0, c->stream, t1);
  if (s->grouped) {
    GroupedArgs gg;
    unsigned gxg = 1;
    gg.xcd_split = ((size_t)na * nb * 8 > (size_t(3) << 20)) ? 1 : 0;
    if (const char* env = std::getenv("SQD_SPMM_XCD")) gg.xcd_split = std::atoi(env) != 0;  // tuning hook
    for (int sp = 0; sp < 2; ++sp) {
      const SpmmGroups& gr = s->groups[sp];
      gg.base[sp] = gr.base.as<int64_t>();
      gg.cnt[sp] = gr.cnt.as<uint32_t>();
      gg.src[sp] = gr.src.as<uint32_t>();
      gg.order[sp] = gr.order.as<uint32_t>();
      gg.coef[sp] = gr.coef.as<double>();
      gg.n[sp] = sp ? nb : na;
      gg.m[sp] = sp ? na : nb;
      gg.ngroups[sp] = (unsigned)gr.ngroups;
      gg.npanels[sp] = (unsigned)((gg.m[sp] + 63) / 64);
      uint64_t blocks;
      if (gg.xcd_split) blocks = 8ull * (((uint64_t)((gg.npanels[sp] + 7) / 8) * (uint64_t)gr.ngroups + 3) / 4);
      else blocks = ((uint64_t)gg.npanels[sp] * (uint64_t)gr.ngroups + 3) / 4;
      gxg = blocks > gxg ? (unsigned)blocks : gxg;
    }
    gg.in[0] = d_c;
    gg.in[1] = s->ct.as<double>();
    gg.out[0] = c->gdense.as<double>();
    gg.out[1] = s->g2t.as<double>();
    gg.stop = c->sigma_stop;
    gg.vec_index = vidx;
    gg.in_stride = in_stride;
    // columns per lane.  Measured (profiles/r05/spmm_gj_probe.txt, HF-centred N x N, us per sigma, 1 | 2 | 4 columns):
    // 1000: 194 | 195 | 283; 2000: 764 | 707 | 793; 3000: 2210 | 2026 | 2301 -- two from ~1000 strings on (a coefficient
    // record through the scalar cache serves twice the multiply-adds; four leave the eight XCDs unevenly loaded)
    int gj_env = 0;
    if (const char* env = std::getenv("SQD_SPMM_GJ")) {  // tuning / test hook
      const int v = std::atoi(env);
      gj_env = (v == 1 || v == 2 || v == 4) ? v : 0;
    }
    const int gj = gj_env ? gj_env : ((na >= 1024 && nb >= 1024) ? 2 : 1);
    gxg = 1;
    for (int sp = 0; sp < 2; ++sp) {
      gg.npanels[sp] = (unsigned)((gg.m[sp] + 64 * gj - 1) / (64 * gj));
      uint64_t blocks;
      if (gg.xcd_split) blocks = 8ull * (((uint64_t)((gg.npanels[sp] + 7) / 8) * (uint64_t)gg.ngroups[sp] + 3) / 4);
      else blocks = ((uint64_t)gg.npanels[sp] * (uint64_t)gg.ngroups[sp] + 3) / 4;
      gxg = blocks > gxg ? (unsigned)blocks : gxg;
    }
    // (measured and dropped, sources under profiles/probes/spmm_rejected/: the group records through an LDS slab shared by
    // four panels -- 117 vs 97 us at 1000^2, 1017 vs 855 at 3000^2 -- and LDS tiles of 128 source rows, 1.4 x slower)
    if (gj == 4) hipLaunchKernelGGL((k_spmm_grouped<4, 2, 4>), dim3(gxg, 2), dim3(256), 0, c->stream, gg);
    else if (gj == 2) hipLaunchKernelGGL((k_spmm_grouped<8, 2, 2>), dim3(gxg, 2), dim3(256), 0, c->stream, gg);
    else hipLaunchKernelGGL((k_spmm_grouped<16, 2, 1>), dim3(gxg, 2), dim3(256), 0, c->stream, gg);
  } else {
  SpmmArgs g;
  unsigned gx = 1;
  // an XCD's L2 (4 MB) holds one panel of the larger side with room to spare: below that the whole vector is L2
  // resident anyway and the tasks are simply dealt out in order
  g.xcd_split = ((size_t)na * nb * 8 > (size_t(3) << 20)) ? 1 : 0;
  if (const char* env = std::getenv("SQD_SPMM_XCD")) g.xcd_split = std::atoi(env) != 0;  // tuning hook
  const int J = spmm_J(na, nb, g.xcd_split != 0);
  for (int sp = 0; sp < 2; ++sp) {
    const SpmmSide& d = s->side[sp];
    g.ptr[sp] = d.ptr.as<int64_t>();
    g.src[sp] = d.src.as<uint32_t>();
    g.val[sp] = d.val.as<double>();
    g.order[sp] = d.order.as<uint32_t>();
    g.n[sp] = d.n;
    g.m[sp] = d.m;
    g.npanels[sp] = (unsigned)((d.m + 64 * J - 1) / (64 * J));
    uint64_t blocks;
    if (g.xcd_split) blocks = 8ull * (((uint64_t)((g.npanels[sp] + 7) / 8) * (uint64_t)d.n + 3) / 4);
    else blocks = ((uint64_t)g.npanels[sp] * (uint64_t)d.n + 3) / 4;
    gx = blocks > gx ? (unsigned)blocks : gx;
  }
  g.in[0] = d_c;
  g.in[1] = s->ct.as<double>();
  g.out[0] = c->gdense.as<double>();
  g.out[1] = s->g2t.as<double>();
  g.stop = c->sigma_stop;
  g.vec_index = vidx;
  g.in_stride = in_stride;
  static const int ru = [] {  // tuning hook: operand rows in flight per wavefront
    const char* env = std::getenv("SQD_SPMM_U");
    return (env && std::atoi(env) == 8) ? 8 : 16;
  }();
  if (J == 4) hipLaunchKernelGGL((k_spmm_rows<4, 8>), dim3(gx, 2), dim3(256), 0, c->stream, g);
  else if (J == 2) hipLaunchKernelGGL((k_spmm_rows<2, 8>), dim3(gx, 2), dim3(256), 0, c->stream, g);
  else if (ru == 8) hipLaunchKernelGGL((k_spmm_rows<1, 8>), dim3(gx, 2), dim3(256), 0, c->stream, g);
  else hipLaunchKernelGGL((k_spmm_rows<1, 16>), dim3(gx, 2), dim3(256), 0, c->stream, g);
  }
  TransArgs t2;
  t2.in = s->g2t.as<double>();
  t2.out = c->gdense.as<double>();
  t2.rows = nb;
  t2.cols = na;
  t2.add = 1;
  t2.stop = c->sigma_stop;
  t2.vec_index = nullptr;
  t2.in_stride = 0;
  hipLaunchKernelGGL(k_spmm_transpose, dim3((unsigned)((na + 63) / 64), (unsigned)((nb + 63) / 64)), dim3(256), 0, c->stream, t2);
  SQD_HIP_CHECK(hipGetLastError());
  return SQD_OK;
}

}  // namespace sqd
