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
#include <cstdlib>
#include <numeric>

#include "sqd_common.h"

namespace sqd {

constexpr int SPMM_U = 8;  // links per round; the merged lists are padded to whole rounds with zero-weight links
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
struct SpmmState {
  SpmmSide side[2];
  DevBuf ct, g2t;  // C^T (nb x na) and H_b C^T (nb x na)
};

void spmm_release(sqd_ctx* c) {
  if (!c->spmm) return;
  SpmmState* s = static_cast<SpmmState*>(c->spmm);
  for (auto& sd : s->side)
    for (DevBuf* b : {&sd.ptr, &sd.src, &sd.val, &sd.order}) b->release();
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
template <int J>
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
  constexpr int U = SPMM_U;
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

// ---- host side
// Is this subspace taken by the sparse-product same-spin path?  (phase 2 of set_subspace; single builds, whole row
// range.)  SQD_SIGMA_SPMM=1 / 0 forces / forbids (the tests run every same-spin formulation on the same inputs).
bool spmm_select(sqd_ctx* c, int64_t na, int64_t nb, int64_t row0, int64_t row1, const int64_t* tot, bool direct) {
  c->sig_spmm = false;
  if (direct || row0 != 0 || row1 != na) return false;
  if (na > 0xfffffffell / 64 || nb > 0xfffffffell / 64) return false;
  const char* env = std::getenv("SQD_SIGMA_SPMM");
  bool on = false;
  if (env) {
    on = std::atoi(env) != 0;
  } else {
    // connected sets from ~1400 strings per spin.  Measured on the MI355X (profiles/r05/connected_probe_spmm.txt,
    // HF-centred N x N, us per sigma, this path | matrix cores | sparse work items): 700: 160 | 111 | 125; 1000: 288 |
    // 236 | 257; 2000: 1100 | 1305 | 1244; 3000: 2899 | 3747 | 5073.  Smaller sets stay with the matrix cores: one
    // launch instead of three, and their blocks are 11-26 % dense.
    const int64_t same_a = tot[0] + tot[1], same_b = tot[2] + tot[3];
    on = na >= 1400 && nb >= 1400 && same_a >= 8 * na && same_b >= 8 * nb;
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
  return SQD_OK;
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
  if (J == 4) hipLaunchKernelGGL(k_spmm_rows<4>, dim3(gx, 2), dim3(256), 0, c->stream, g);
  else if (J == 2) hipLaunchKernelGGL(k_spmm_rows<2>, dim3(gx, 2), dim3(256), 0, c->stream, g);
  else hipLaunchKernelGGL(k_spmm_rows<1>, dim3(gx, 2), dim3(256), 0, c->stream, g);
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
