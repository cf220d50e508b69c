// Device-resident single-root Davidson for P (H + penalty) P.
//
// Replaces pyscf kernel_fixed_space -> FCISolver.eig -> lib.davidson1 (numpy BLAS-1 on the host)
// as called from qiskit_addon_sqd/fermion.py:721-723 and :810-818.  Control flow and constants
// follow SURVEY.md Appendix A.6: tol on |dE| and sqrt(tol) on |r|, preconditioner
// r/(hdiag - e + 1e-4), Gram-Schmidt against the whole basis, lindep drop, restart when the basis
// reaches max_space.  One deliberate difference: at a restart pyscf throws the fresh correction
// vector away and spends a sigma build on the Ritz vector; here the basis collapses to
// {Ritz vector, correction} and A*Ritz is formed by linear combination, saving that sigma build.
//
// All vectors (basis X, A X, hdiag) stay in HBM; only the (m x m) projected matrix and a handful of
// norms cross to the host per iteration.  BLAS-1 work is fused: one pass builds residual +
// preconditioned correction + its overlaps with the basis; reductions are fixed-order (bitwise
// reproducible run to run).
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "sqd_common.h"
#include "sqd_device.h"

namespace sqd {

constexpr int NV = 16;       // vectors per fused reduction launch
constexpr int RED_BLOCKS = 512;
constexpr int RED_T = 512;    // 8 waves per workgroup: half as many partials to fold as with 256

// ---- loads over "the first nvec of up to N vectors" WITHOUT a branch per vector.  Written as
// `if (v < nvec) acc += X[v*stride+i] * y` the compiler emits, per vector, a scalar branch around
// {global_load; s_waitcnt vmcnt(0); v_fmac}: nvec dependent memory round trips in sequence, which was the
// whole duration of the Davidson BLAS-1 kernels (ISA + kernel trace: 9-21 us growing with the basis size).
// Here every slot is loaded -- slots past nvec re-read vector 0 (an L1 hit on a line already requested) --
// so all loads are in flight together, and the consumers select instead of branching.
template <int N>
__device__ inline void load_vectors(const double* __restrict__ X, int64_t stride, int nvec, int64_t i, double (&out)[N]) {
#pragma unroll
  for (int v = 0; v < N; ++v) out[v] = X[(int64_t)(v < nvec ? v : 0) * stride + i];
}
// the same for one row of a [blocks][width] partial-sum array (ordinary / device-coherent loads)
template <int N, bool COHERENT>
__device__ inline void load_partials(const double* partial, int64_t row_offset, int nv, double (&out)[N]) {
#pragma unroll
  for (int v = 0; v < N; ++v) {
    const double* p = &partial[row_offset + (v < nv ? v : 0)];
    out[v] = COHERENT ? coherent_load(p) : *p;
  }
}

// partial[block*NV + v] = sum_i X[v*stride + i] * y[i]   (v < nvec <= NV)
__global__ void k_dots(int64_t n, const double* __restrict__ X, int64_t stride, int nvec,
                       const double* __restrict__ y, double* __restrict__ partial) {
  __shared__ double red[16 * NV];
  double acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double yv = y[i];
    double xv[NV];
    load_vectors<NV>(X, stride, nvec, i, xv);
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] += (v < nvec) ? xv[v] * yv : 0.0;
  }
  block_sum_multi<NV>(acc, nvec, red);
  if ((int)threadIdx.x < nvec)
    partial[(int64_t)blockIdx.x * NV + threadIdx.x] = block_sum_multi_get<NV>(red, threadIdx.x);
}

struct Coef {
  double v[SQD_MAX_SPACE + 2];
};

// out = sum_{v<nvec} coef[v] * X[v]
__global__ void k_lincomb(int64_t n, const double* __restrict__ X, int64_t stride, int nvec, const Coef coef,
                          double* __restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double s = 0.0;
    for (int v0 = 0; v0 < nvec; v0 += 8) {  // eight vectors' loads in flight per round
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = X[(int64_t)(v0 + u < nvec ? v0 + u : v0) * stride + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (v0 + u < nvec) ? coef.v[v0 + u < nvec ? v0 + u : v0] * x[u] : 0.0;
    }
    out[i] = s;
  }
}

// r = sum_v coef[v] (AX_v - e X_v);  t = r / (hdiag - e + 1e-4);  t stored to `out`.
// partial[block*width + {0: |r|^2, 1: |t|^2, 2+v: X_v . t}]
// Diagonal of the spin penalty for the preconditioner (pyscf leaves hdiag un-shifted, which makes
// the (S^2-ss)^2 form crawl; the operator is unchanged, only the preconditioner is better):
//   form 1: shift (d - ss);  form 2: shift ((d - ss)^2 + n_flip),  d = sz(sz+1) + |B\A|, n_flip = |B\A||A\B|
struct PenaltyDiag {
  int form;
  double shift, ss, szterm;
  const uint64_t* sa;
  const uint64_t* sb;
  int64_t nb;
};
__device__ inline double penalty_diag(const PenaltyDiag& p, int64_t i) {
  if (p.form == 0) return 0.0;
  const uint64_t A = p.sa[i / p.nb], B = p.sb[i % p.nb];
  const double nba = (double)__popcll(B & ~A);
  const double d = p.szterm + nba - p.ss;
  if (p.form == 1) return p.shift * d;
  return p.shift * (d * d + nba * (double)__popcll(A & ~B));
}

// When the residual totals say the solve is over -- (|dE| < tol and |r|^2 < tol2) or a vanishing residual /
// correction -- the device raises *flag itself, so that work the host enqueued ahead (next sigma, ...) returns
// at once.  The host applies the SAME comparisons to the same numbers.  flag == nullptr: no rule.
struct StopRule {
  int* flag;
  int de_small;
  double tol2, lindep;
};
template <int N>
__device__ inline void finish_and_post(const double* partial, int width, int nv, unsigned* counter,
                                       double* __restrict__ dsums, double* mail, long long seq, double* red,
                                       const StopRule rule);

template <int MV>
__global__ void k_residual_precond(int64_t n, const double* __restrict__ X, const double* __restrict__ AX,
                                   int64_t stride, int nvec, const Coef coef, double e,
                                   const double* __restrict__ hdiag, const PenaltyDiag pd, double* __restrict__ out,
                                   double* __restrict__ partial, int width) {
  // vals[0] = |r|^2, vals[1] = |t|^2, vals[2+v] = X_v . t ; MV bounds the basis size (registers)
  __shared__ double red[16 * (MV + 2)];
  double vals[MV + 2];
#pragma unroll
  for (int v = 0; v < MV + 2; ++v) vals[v] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double r = 0.0;
    double xv[MV];
    const double hd = hdiag[i];
    // eight (X_v, AX_v) pairs requested per round, branch-free (see load_vectors); X_v is kept for the overlaps
#pragma unroll
    for (int v0 = 0; v0 < MV; v0 += 8) {
      double a8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v0 + u < MV) {
          const int64_t off = (int64_t)(v0 + u < nvec ? v0 + u : 0) * stride + i;
          xv[v0 + u] = X[off];
          a8[u] = AX[off];
        }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v0 + u < MV) r += (v0 + u < nvec) ? coef.v[v0 + u] * (a8[u] - e * xv[v0 + u]) : 0.0;
    }
    const double t = r / (hd + penalty_diag(pd, i) - e + 1e-4);
    out[i] = t;
    vals[0] += r * r;
    vals[1] += t * t;
#pragma unroll
    for (int v = 0; v < MV; ++v) vals[2 + v] += (v < nvec) ? xv[v] * t : 0.0;
  }
  block_sum_multi<MV + 2>(vals, nvec + 2, red);
  // per-workgroup partials only: k_orth_dev (next in the stream) folds them
  if ((int)threadIdx.x < nvec + 2)
    partial[(int64_t)blockIdx.x * width + threadIdx.x] = block_sum_multi_get<MV + 2>(red, threadIdx.x);
}

// per-block (min value, index) over hdiag; tril != 0 restricts to A >= B (pyscf _get_init_guess
// when nelec_a == nelec_b and na == nb)
__global__ void k_argmin(int64_t n, int64_t nb, int tril_only, const double* __restrict__ h,
                         double* __restrict__ pmin, int64_t* __restrict__ pidx) {
  double best = 1e300;
  int64_t bi = -1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (tril_only && (i / nb) < (i % nb)) continue;
    const double v = h[i];
    if (v < best || (v == best && i < bi)) {
      best = v;
      bi = i;
    }
  }
  block_argmin(best, bi);
  if (threadIdx.x == 0) {
    pmin[blockIdx.x] = best;
    pidx[blockIdx.x] = bi;
  }
}

// pyscf get_init_guess: unit vector at addr, +1e-5 on the first and -1e-5 on the last element,
// normalised here with the closed-form norm (no reduction, no host round trip)
// (every workgroup repeats the final stage of the argmin over the <= RED_BLOCKS per-block candidates
// instead of a separate single-workgroup launch)
__global__ void k_init_guess(int64_t n, const double* __restrict__ pmin, const int64_t* __restrict__ pidx, int nblocks,
                             double* __restrict__ x) {
  __shared__ long long s_addr;
  {
    double best = 1e300;
    int64_t bi = -1;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
      const double v = pmin[b];
      const int64_t i = pidx[b];
      if (i >= 0 && (v < best || (v == best && i < bi) || bi < 0)) {
        best = v;
        bi = i;
      }
    }
    block_argmin(best, bi);
    if (threadIdx.x == 0) s_addr = bi < 0 ? 0 : bi;
    __syncthreads();
  }
  const int64_t addr = s_addr;
  auto f = [=](int64_t i) { return ((i == addr) ? 1.0 : 0.0) + ((i == 0) ? 1e-5 : 0.0) - ((i == n - 1) ? 1e-5 : 0.0); };
  double nn = f(0) * f(0);
  if (n - 1 != 0) nn += f(n - 1) * f(n - 1);
  if (addr != 0 && addr != n - 1) nn += f(addr) * f(addr);
  const double inv = 1.0 / sqrt(nn);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    x[i] = f(i) * inv;
}

// ------------------------------------------------------------------ host helpers
static inline unsigned red_blocks(int64_t n) {
  int64_t b = (n + RED_T - 1) / RED_T;  // one element per thread while the grid lasts: latency, not bandwidth
  if (b > RED_BLOCKS) b = RED_BLOCKS;
  static const int64_t cap = [] {  // tuning hook
    const char* env = std::getenv("SQD_RED_BLOCKS");
    return env ? (int64_t)std::atoi(env) : (int64_t)0;
  }();
  if (cap > 0 && b > cap) b = cap;
  if (b < 1) b = 1;
  return (unsigned)b;
}

// Device scalar block: scal[0] = 1/|t'| (0 if linearly dependent), scal[1] = |t'|^2 of the last
// orthogonalisation, scal[2..] = outputs of the latest reduction.
constexpr int SCAL_RED = 2;

// Final stage of a reduction + hand-over to the host in ONE single-workgroup kernel: column sums of
// the partial array (fixed order), written with scal[0..2) straight into the host-visible mailbox,
// then a system-scope fence and the sequence word.  The host spins on that word (bounded) -- no copy
// engine, no stream synchronisation on the critical path.
constexpr int MAIL_PAYLOAD = 8;  // doubles; mail[0] is the sequence word
template <int MAXV>
__global__ void k_reduce_to_mail(const double* __restrict__ partial, int nblocks, int width, int nv,
                                 const double* __restrict__ scal, double* __restrict__ mail, long long seq) {
  __shared__ double red[16 * MAXV];
  double vals[MAXV];
#pragma unroll
  for (int v = 0; v < MAXV; ++v) vals[v] = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    double p[MAXV];
    load_partials<MAXV, false>(partial, (int64_t)b * width, nv, p);
#pragma unroll
    for (int v = 0; v < MAXV; ++v) vals[v] += (v < nv) ? p[v] : 0.0;
  }
  block_sum_multi<MAXV>(vals, nv, red);
  if ((int)threadIdx.x < nv) mail_store(&mail[MAIL_PAYLOAD + SCAL_RED + threadIdx.x], block_sum_multi_get<MAXV>(red, threadIdx.x));
  if (threadIdx.x == 0) {
    mail_store(&mail[MAIL_PAYLOAD + 0], scal[0]);
    mail_store(&mail[MAIL_PAYLOAD + 1], scal[1]);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    *reinterpret_cast<volatile long long*>(mail) = seq;
  }
}

// ---- fused reductions for the Davidson loop: the workgroup that arrives LAST folds the per-block
// partials (fixed order => bitwise reproducible), leaves the totals on the device (dsums, for the
// kernels that follow in the stream) and posts them to a host-visible mailbox.  No single-workgroup
// reduction launch per hand-over, and the host is not needed between producer and consumer kernels.
constexpr int MAIL_SLOT = 128;  // doubles per mailbox slot (slot 0: projected-matrix column, slot 1: residual)
constexpr unsigned COUNT_GROUPS = 16;  // arrival counters: word 0 = groups done, words 1..16 = per group
constexpr unsigned COUNT_STRIDE = 32;  // ... each in its own 128-byte line: atomics on one line serialise in one L2 channel
template <int N>
__device__ inline void finish_and_post(const double* partial, int width, int nv, unsigned* counter,
                                       double* __restrict__ dsums, double* mail, long long seq, double* red,
                                       const StopRule rule) {
  __shared__ int s_last;
  // the callers wrote their partials with coherent_store: once those stores have completed (waitcnt) the
  // workgroup may be counted; no L2-wide fence (see sqd_device.h)
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    // two-level arrival count (COUNT_GROUPS group words + a top word): device-scope atomics on ONE word
    // serialise at ~50 ns each, which for a few hundred workgroups costs more than the reduction itself
    int last = 0;
    const unsigned G = COUNT_GROUPS, grp = blockIdx.x % G;
    const unsigned gsize = (gridDim.x - grp + G - 1) / G, ngroups = gridDim.x < G ? gridDim.x : G;
    if (atomicAdd(&counter[COUNT_STRIDE * (1 + grp)], 1u) == gsize - 1) {
      atomicExch(&counter[COUNT_STRIDE * (1 + grp)], 0u);  // ready for the next fused reduction on this stream
      if (atomicAdd(&counter[0], 1u) == ngroups - 1) {
        atomicExch(&counter[0], 0u);
        last = 1;
      }
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  double vals[N];
#pragma unroll
  for (int v = 0; v < N; ++v) vals[v] = 0.0;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) {
    double p[N];
    load_partials<N, true>(partial, (int64_t)b * width, nv, p);  // all requests in flight together
#pragma unroll
    for (int v = 0; v < N; ++v) vals[v] += (v < nv) ? p[v] : 0.0;
  }
  block_sum_multi<N>(vals, nv, red);
  if ((int)threadIdx.x < nv) {
    const double s = block_sum_multi_get<N>(red, threadIdx.x);
    dsums[threadIdx.x] = s;
    mail_store(&mail[MAIL_PAYLOAD + SCAL_RED + threadIdx.x], s);
  }
  if (rule.flag && threadIdx.x == 0) {
    const double s0 = block_sum_multi_get<N>(red, 0), s1 = block_sum_multi_get<N>(red, 1);
    if ((rule.de_small && s0 < rule.tol2) || !(s0 > rule.lindep) || !(s1 > 0.0)) *rule.flag = 1;
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    *reinterpret_cast<volatile long long*>(mail) = seq;
  }
}

// sums[0] = |X_{nvec-1}|^2, sums[1+v] = X_v . y  (v < nvec <= MV); y = A X_{nvec-1} in the Davidson loop
template <int MV>
__global__ void k_dots_post(int64_t n, const double* __restrict__ X, int64_t stride, int nvec,
                            const double* __restrict__ y, double* __restrict__ partial, int width, unsigned* counter,
                            double* __restrict__ dsums, double* mail, long long seq, const int* stop) {
  __shared__ double red[16 * (MV + 1)];
  if (stop && *stop) return;  // enqueued ahead of a residual that ended the solve: nobody waits for this
  double acc[MV + 1];
#pragma unroll
  for (int v = 0; v < MV + 1; ++v) acc[v] = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const double yv = y[i];
    double xv[MV];
    load_vectors<MV>(X, stride, nvec, i, xv);
#pragma unroll
    for (int v = 0; v < MV; ++v) {
      acc[1 + v] += (v < nvec) ? xv[v] * yv : 0.0;
      acc[0] += (v == nvec - 1) ? xv[v] * xv[v] : 0.0;
    }
  }
  block_sum_multi<MV + 1>(acc, nvec + 1, red);
  if ((int)threadIdx.x < nvec + 1)
    coherent_store(&partial[(int64_t)blockIdx.x * width + threadIdx.x], block_sum_multi_get<MV + 1>(red, threadIdx.x));
  finish_and_post<MV + 1>(partial, width, nvec + 1, counter, dsums, mail, seq, red, StopRule{nullptr, 0, 0.0, 0.0});
}

// t <- scale * t - sum_v g_v X_v with everything derived on the device from the residual kernel's totals
// (dsums = {|r|^2, |t|^2, X_v . t}) and the per-vector normalisation factors sv (basis vector v is
// sv_v * X_v): g'_v = sv_v (X_v . t) / |t|, c2 = sum g'_v^2.  The basis is orthonormal, so
// |t/|t| - sum g'_v sv_v X_v|^2 = 1 - c2 is known before the vector is formed: when 1 - c2 > 1e-3 the
// result is normalised in the same pass; otherwise it is left with its true (small) norm.  Either way the
// next k_dots_post measures |X_new|^2 and the host carries 1/sqrt of it as sv_new, so no separate
// normalisation pass and no host decision is needed here.
//
// The residual kernel leaves only per-workgroup partials: EVERY workgroup here folds them itself (fixed
// order, the same arithmetic everywhere => the same totals to the bit), which is cheaper than a finishing
// step inside the residual kernel (arrival atomics + coherent re-read, ~8 us) and needs no extra launch.
// Workgroup 0 posts the totals to the host mailbox and applies the stop rule for the kernels enqueued ahead;
// every workgroup applies it to itself.
template <int MV>
__global__ void k_orth_dev(int64_t n, const double* __restrict__ X, int64_t stride, int nvec, const Coef sv,
                           const double* __restrict__ partial, int nblocks, int width, double* __restrict__ t,
                           double* mail, long long seq, const StopRule rule, const int* stop) {
  __shared__ double red[16 * (MV + 2)];
  __shared__ double tot[MV + 2];
  __shared__ double g[MV + 2];
  __shared__ double s_scale;
  __shared__ int s_stop;
  if (stop && *stop) return;
  const int nv = nvec + 2;
  {
    double vals[MV + 2];
#pragma unroll
    for (int v = 0; v < MV + 2; ++v) vals[v] = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
      double p[MV + 2];
      load_partials<MV + 2, false>(partial, (int64_t)b * width, nv, p);
#pragma unroll
      for (int v = 0; v < MV + 2; ++v) vals[v] += (v < nv) ? p[v] : 0.0;
    }
    block_sum_multi<MV + 2>(vals, nv, red);
    if ((int)threadIdx.x < nv) tot[threadIdx.x] = block_sum_multi_get<MV + 2>(red, threadIdx.x);
  }
  __syncthreads();
  const double rr = tot[0], tt = tot[1];
  if ((int)threadIdx.x < nvec) g[threadIdx.x] = (tt > 0.0) ? sv.v[threadIdx.x] * tot[2 + threadIdx.x] / sqrt(tt) : 0.0;
  __syncthreads();
  if (threadIdx.x == 0) {
    double c2 = 0.0;
    for (int v = 0; v < nvec; ++v) c2 += g[v] * g[v];
    const double inv = (1.0 - c2 > 1e-3) ? 1.0 / sqrt(1.0 - c2) : 1.0;
    s_scale = (tt > 0.0) ? inv / sqrt(tt) : 0.0;
    for (int v = 0; v < nvec; ++v) g[v] *= inv * sv.v[v];
    s_stop = ((rule.de_small && rr < rule.tol2) || !(rr > rule.lindep) || !(tt > 0.0)) ? 1 : 0;
  }
  if (blockIdx.x == 0) {
    if ((int)threadIdx.x < nv) mail_store(&mail[MAIL_PAYLOAD + SCAL_RED + threadIdx.x], tot[threadIdx.x]);
    __builtin_amdgcn_s_waitcnt(0);
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (s_stop && rule.flag) *rule.flag = 1;
    __threadfence_system();
    *reinterpret_cast<volatile long long*>(mail) = seq;
  }
  if (s_stop) return;  // the correction is not needed (and may be 0/0)
  const double scale = s_scale;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    double s = scale * t[i];
    for (int v0 = 0; v0 < nvec; v0 += 8) {  // eight vectors' loads in flight per round
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = X[(int64_t)(v0 + u < nvec ? v0 + u : v0) * stride + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s -= (v0 + u < nvec) ? g[v0 + u < nvec ? v0 + u : v0] * x[u] : 0.0;
    }
    t[i] = s;
  }
}

// host side of finish_and_post: wait for sequence word `seq` in mailbox slot `slot`, read nv totals
static int wait_mail(sqd_ctx* c, int slot, long long seq, int nv, double* sums) {
  const double* mail = c->h_mail + (size_t)slot * MAIL_SLOT;
  volatile const long long* flag = reinterpret_cast<volatile const long long*>(mail);
  bool seen = false;
  for (long spin = 0; spin < 20000000L; ++spin) {
    if (*flag == seq) {
      seen = true;
      break;
    }
    __builtin_ia32_pause();
  }
  if (!seen) {  // fall back to a plain synchronisation (also surfaces asynchronous kernel errors)
    SQD_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (*flag != seq) {
      set_error("device mailbox was not written");
      return SQD_ERR_HIP;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  for (int v = 0; v < nv; ++v) sums[v] = mail[MAIL_PAYLOAD + SCAL_RED + v];
  return SQD_OK;
}

// sums[0..nv) = column sums of the device partial array; c->h_pinned[0..2) = scal[0..2).
static int fetch_sums(sqd_ctx* c, int nblocks, int width, int nv, double* sums) {
  const long long seq = ++c->mail_seq;
  if (nv <= 16)
    hipLaunchKernelGGL((k_reduce_to_mail<16>), dim3(1), dim3(128), 0, c->stream, (const double*)c->partial.as<double>(),
                       nblocks, width, nv, (const double*)c->scal.as<double>(), c->d_mail, seq);
  else
    hipLaunchKernelGGL((k_reduce_to_mail<SQD_MAX_SPACE + 4>), dim3(1), dim3(128), 0, c->stream,
                       (const double*)c->partial.as<double>(), nblocks, width, nv, (const double*)c->scal.as<double>(),
                       c->d_mail, seq);
  SQD_HIP_CHECK(hipGetLastError());
  volatile long long* flag = reinterpret_cast<volatile long long*>(c->h_mail);
  bool seen = false;
  for (long spin = 0; spin < 20000000L; ++spin) {
    if (*flag == seq) {
      seen = true;
      break;
    }
    __builtin_ia32_pause();
  }
  if (!seen) {  // fall back to a plain synchronisation (also surfaces asynchronous kernel errors)
    SQD_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (*flag != seq) {
      set_error("device mailbox was not written");
      return SQD_ERR_HIP;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  c->h_pinned[0] = c->h_mail[MAIL_PAYLOAD + 0];
  c->h_pinned[1] = c->h_mail[MAIL_PAYLOAD + 1];
  for (int v = 0; v < nv; ++v) sums[v] = c->h_mail[MAIL_PAYLOAD + SCAL_RED + v];
  return SQD_OK;
}

// dots[v] = X_v . y for v < nvec (any nvec <= SQD_MAX_SPACE+1), fixed-order reduction
static int multi_dot(sqd_ctx* c, const double* X, int64_t stride, int nvec, const double* y, double* dots) {
  const int64_t n = c->D;
  const unsigned nb = red_blocks(n);
  for (int v0 = 0; v0 < nvec; v0 += NV) {
    const int nv = (nvec - v0 < NV) ? (nvec - v0) : NV;
    hipLaunchKernelGGL(k_dots, dim3(nb), dim3(RED_T), 0, c->stream, n, X + (int64_t)v0 * stride, stride, nv, y,
                       c->partial.as<double>());
    SQD_HIP_CHECK(hipGetLastError());
    SQD_TRY(fetch_sums(c, (int)nb, NV, nv, dots + v0));
  }
  return SQD_OK;
}

int dev_dot(sqd_ctx* c, const double* x, const double* y, double* out) {
  SQD_TRY(c->partial.reserve((size_t)2 * RED_BLOCKS * (SQD_MAX_SPACE + 4) * 8));
  SQD_TRY(c->scal.reserve(8192));
  return multi_dot(c, x, 0, 1, y, out);
}

// cyclic Jacobi for a small symmetric matrix; eigenvalues ascending in w, eigenvectors in columns of V
static void jacobi_eigh(int n, const double* Ain, double* w, double* V) {
  std::vector<double> A(Ain, Ain + n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, dia = 0.0;
    for (int i = 0; i < n; ++i) {
      dia += A[i * n + i] * A[i * n + i];
      for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
    }
    // converged to working precision (quadratic convergence: one more sweep would change nothing)
    if (off <= 1e-32 * dia || off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
        for (int k = 0; k < n; ++k) {
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = cs * akp - sn * akq;
          A[k * n + q] = sn * akp + cs * akq;
        }
        for (int k = 0; k < n; ++k) {
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = cs * apk - sn * aqk;
          A[q * n + k] = sn * apk + cs * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = cs * vkp - sn * vkq;
          V[k * n + q] = sn * vkp + cs * vkq;
        }
      }
  }
  std::vector<int> order(n);
  for (int i = 0; i < n; ++i) order[i] = i;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j)
      if (A[order[j] * n + order[j]] < A[order[i] * n + order[i]]) std::swap(order[i], order[j]);
  std::vector<double> Vs(n * n);
  for (int j = 0; j < n; ++j) {
    w[j] = A[order[j] * n + order[j]];
    for (int k = 0; k < n; ++k) Vs[k * n + j] = V[k * n + order[j]];
  }
  std::memcpy(V, Vs.data(), sizeof(double) * n * n);
}

// Lowest eigenpair of the projected matrix after the basis grew by ONE vector, by Rayleigh-quotient iteration
// from the previous Ritz vector padded with a zero (a start whose residual is already small).  The previous
// matrix is the leading principal block of this one, so by Cauchy interlacing  l1(new) <= l1(old) <= l2(new):
// an eigenvalue found at or below the old Ritz value IS the lowest one -- that test, plus a residual at
// rounding level, is the acceptance rule; anything else returns false and the caller runs the Jacobi solver.
// Cost ~2 LU factorisations of an m x m matrix (1 us at m = 12) instead of 13 us of cold Jacobi sweeps.
static bool lowest_eig_rqi(int n, const double* A, const double* v_old, double e_old, double* e_out, double* v_out) {
  if (n < 2 || n > SQD_MAX_SPACE + 1) return false;
  double x[SQD_MAX_SPACE + 2], y[SQD_MAX_SPACE + 2], M[(SQD_MAX_SPACE + 1) * (SQD_MAX_SPACE + 1)];
  double nrm = 0.0, anorm = 0.0;
  for (int i = 0; i < n; ++i) {
    x[i] = (i < n - 1) ? v_old[i] : 0.0;
    nrm += x[i] * x[i];
    double r = 0.0;
    for (int j = 0; j < n; ++j) r += std::fabs(A[i * n + j]);
    anorm = r > anorm ? r : anorm;
  }
  if (!(nrm > 0.0) || !(anorm > 0.0)) return false;
  nrm = 1.0 / std::sqrt(nrm);
  for (int i = 0; i < n; ++i) x[i] *= nrm;
  auto rayleigh = [&](const double* v, double* Av) {
    double t = 0.0;
    for (int i = 0; i < n; ++i) {
      double r = 0.0;
      for (int j = 0; j < n; ++j) r += A[i * n + j] * v[j];
      Av[i] = r;
      t += v[i] * r;
    }
    return t;
  };
  double theta = rayleigh(x, y);
  const double tiny = 2.3e-16 * anorm;
  for (int it = 0; it < 6; ++it) {
    // residual of the current pair
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += (y[i] - theta * x[i]) * (y[i] - theta * x[i]);
    if (std::sqrt(res) <= 8.0 * tiny) {
      if (!(theta <= e_old + 64.0 * tiny)) return false;  // not provably the lowest eigenvalue
      // (a new vector that does not couple to the old Ritz vector leaves that pair an eigenpair of the grown
      // matrix although its own diagonal may lie lower: the lowest eigenvalue is below every diagonal element)
      for (int i = 0; i < n; ++i)
        if (theta > A[i * n + i] + 64.0 * tiny) return false;
      *e_out = theta;
      for (int i = 0; i < n; ++i) v_out[i] = x[i];
      return true;
    }
    // y = (A - theta I)^-1 x   (Gaussian elimination, partial pivoting; a vanishing pivot is what converges it)
    for (int i = 0; i < n * n; ++i) M[i] = A[i];
    for (int i = 0; i < n; ++i) {
      M[i * n + i] -= theta;
      y[i] = x[i];
    }
    for (int k = 0; k < n; ++k) {
      int p = k;
      for (int i = k + 1; i < n; ++i)
        if (std::fabs(M[i * n + k]) > std::fabs(M[p * n + k])) p = i;
      if (p != k) {
        for (int j = 0; j < n; ++j) std::swap(M[k * n + j], M[p * n + j]);
        std::swap(y[k], y[p]);
      }
      if (std::fabs(M[k * n + k]) < tiny) M[k * n + k] = (M[k * n + k] < 0.0) ? -tiny : tiny;
      const double inv = 1.0 / M[k * n + k];
      for (int i = k + 1; i < n; ++i) {
        const double f = M[i * n + k] * inv;
        if (f == 0.0) continue;
        for (int j = k + 1; j < n; ++j) M[i * n + j] -= f * M[k * n + j];
        y[i] -= f * y[k];
      }
    }
    for (int i = n - 1; i >= 0; --i) {
      double r = y[i];
      for (int j = i + 1; j < n; ++j) r -= M[i * n + j] * y[j];
      y[i] = r / M[i * n + i];
    }
    double yn = 0.0, dot = 0.0;
    for (int i = 0; i < n; ++i) {
      yn += y[i] * y[i];
      dot += y[i] * x[i];
    }
    if (!(yn > 0.0) || !std::isfinite(yn)) return false;
    yn = ((dot < 0.0) ? -1.0 : 1.0) / std::sqrt(yn);  // keep the orientation of the previous Ritz vector
    for (int i = 0; i < n; ++i) x[i] = y[i] * yn;
    theta = rayleigh(x, y);
  }
  return false;
}

// pyscf get_init_guess (direct_spin1._get_init_guess): unit vector at the lowest diagonal element -- searched over
// the lower triangle A >= B when nelec_a == nelec_b and na == nb -- plus the +-1e-5 noise, normalised
int enqueue_init_guess(sqd_ctx* c, double* x) {
  const int64_t D = c->D;
  const unsigned gb = red_blocks(D);
  SQD_TRY(c->partial.reserve((size_t)2 * RED_BLOCKS * (SQD_MAX_SPACE + 4) * 8));
  const int tril_only = (c->nelec[0] == c->nelec[1] && c->na == c->nb) ? 1 : 0;
  double* pmin = c->partial.as<double>();
  int64_t* pidx = reinterpret_cast<int64_t*>(pmin + RED_BLOCKS);
  hipLaunchKernelGGL(k_argmin, dim3(gb), dim3(RED_T), 0, c->stream, D, c->nb, tril_only,
                     (const double*)c->hdiag.as<double>(), pmin, pidx);
  hipLaunchKernelGGL(k_init_guess, dim3(gb), dim3(RED_T), 0, c->stream, D, (const double*)pmin, (const int64_t*)pidx,
                     (int)gb, x);
  SQD_HIP_CHECK(hipGetLastError());
  return SQD_OK;
}

int run_davidson(sqd_ctx* c, const sqd_davidson_opts* o, const double* ci0_host, sqd_davidson_stats* st,
                 bool defer_sync) {
  if (!c->have_subspace) {
    set_error("no subspace set");
    return SQD_ERR_STATE;
  }
  const int64_t D = c->D;
  int max_space = o->max_space;
  if (max_space < 2) max_space = 2;
  if (max_space > SQD_MAX_SPACE) max_space = SQD_MAX_SPACE;
  const double tol = o->tol, lindep = o->lindep;
  // residual threshold: sqrt(tol)/32 by default (pyscf: sqrt(tol)).  Energies would be fine with pyscf's
  // value (second order in the residual without a penalty), but the orbital occupancies that steer the next
  // configuration-recovery round are FIRST order in it: 1e-4 with pyscf's threshold, 1e-6 with this one, and
  // a seeded SQD run only reproduces if they are stable.  tol_residual = sqrt(tol) restores pyscf's rule
  // (HF-centred headline: 29 instead of 40 sigma builds, <c|H|c> 3e-10 Ha off).
  const double toloose = (o->tol_residual > 0.0) ? o->tol_residual : std::sqrt(o->tol) / 32.0;
  hipStream_t s = c->stream;
  const int nvecs = max_space + 1;
  SQD_TRY(c->X.reserve((size_t)nvecs * D * 8));
  SQD_TRY(c->AX.reserve((size_t)nvecs * D * 8));
  SQD_TRY(c->sol.reserve((size_t)D * 8));
  SQD_TRY(c->partial.reserve((size_t)2 * RED_BLOCKS * (SQD_MAX_SPACE + 4) * 8));
  SQD_TRY(c->scal.reserve(8192));
  double* X = c->X.as<double>();
  double* AX = c->AX.as<double>();
  const unsigned gb = red_blocks(D);
  const int width = SQD_MAX_SPACE + 4;

  SQD_HIP_CHECK(hipEventRecord(c->ev[2], s));
  // ---- initial vector
  if (ci0_host) {
    SQD_HIP_CHECK(hipMemcpyAsync(X, ci0_host, D * 8, hipMemcpyHostToDevice, s));
  } else {
    SQD_TRY(enqueue_init_guess(c, X));
  }
  // (a user vector is not normalised on the device: the first fused reduction measures |X_0|^2 and the
  // factor is carried in sv like that of every later basis vector)
  double* scal = c->scal.as<double>();
  double* dsums_col = scal + 8;    // totals of the latest k_dots_post

  constexpr int COUNT_DOUBLES = (int)((COUNT_GROUPS + 1) * COUNT_STRIDE * sizeof(unsigned) / sizeof(double));
  unsigned* counter = reinterpret_cast<unsigned*>(scal + 128);
  int* stop_flag = reinterpret_cast<int*>(scal + 128 + COUNT_DOUBLES);
  SQD_HIP_CHECK(hipMemsetAsync(counter, 0, (COUNT_DOUBLES + 1) * sizeof(double), s));  // arrival counters and the stop flag
  const double tol2 = toloose * toloose;
  double* mail_col = c->d_mail;
  double* mail_res = c->d_mail + MAIL_SLOT;

  PenaltyDiag pd;
  {
    int form = o->use_spin;
    const double szh = 0.5 * std::abs(c->nelec[0] - c->nelec[1]);
    if (form == 3) form = (o->ss < szh * (szh + 1.0) + 0.1) ? 1 : 2;
    const double sz = 0.5 * (c->nelec[0] - c->nelec[1]);
    pd.form = form;
    pd.shift = o->shift;
    pd.ss = o->ss;
    pd.szterm = sz * (sz + 1.0);
    pd.sa = c->sp[0].strs.as<uint64_t>();
    pd.sb = c->sp[1].strs.as<uint64_t>();
    pd.nb = c->nb;
  }
  std::vector<double> heff((size_t)nvecs * nvecs, 0.0), sub, w(nvecs), V((size_t)nvecs * nvecs);
  std::vector<double> sums(width);
  std::vector<double> v_eig(nvecs + 1, 0.0), v_new(nvecs + 1, 0.0);  // lowest Ritz vector of the last projected problem
  int m_eig = -1;                                                     // ... and that problem's size
  static const bool use_rqi = std::getenv("SQD_EIG_JACOBI") == nullptr;  // A/B hook: always the Jacobi solver
  // Pipelined loop.  Per iteration the stream holds
  //   sigma(X_new) -> k_dots_post [mailbox 0] -> (host: small eigenproblem) -> k_residual_precond [mailbox 1]
  //   -> k_orth_dev -> (restart kernels)
  // and the host enqueues the NEXT sigma + k_dots_post before it looks at mailbox 1: the residual norm only
  // decides whether to stop, so the decision overlaps with the next sigma, and the one sigma in flight when
  // the solve converges is discarded (never counted).  One host stall per iteration instead of two.
  // Basis vector v is sv_v * X_v (sv_v = 1/|X_v|, measured by k_dots_post): normalisation never costs a pass.
  Coef coef, raw, sv;
  int m = 1;  // basis size; X[m-1] is the newest vector, its sigma not yet built
  int mc = 1; // number of basis vectors the current Ritz coefficients refer to
  coef.v[0] = 1.0;
  sv.v[0] = 1.0;
  double e = 0.0, elast = 0.0, rnorm = 0.0, de = 0.0;
  bool conv = false, have_res = false, stop = false;
  long long seq_res = 0;
  int m_res = 0;  // basis size of the residual kernel whose mailbox is outstanding
  int nsig = 0, nev = 0, it = 0;
  // every time_sigma_every-th sigma launch of this context is bracketed by events (stats->ms_sigma); an
  // event pair costs ~10 us of stream time, so this is sampling, and off unless asked for
  const int ev_every = o->time_sigma_every > 0 ? o->time_sigma_every : 0;
  const int max_ev = ev_every ? (int)c->sig_ev.size() / 3 : 0;
  bool first = true;
  // outcome of the outstanding residual hand-over: sets rnorm, conv, stop
  auto settle_residual = [&]() -> int {
    have_res = false;
    SQD_TRY(wait_mail(c, 1, seq_res, m_res + 2, sums.data()));
    rnorm = std::sqrt(sums[0]);
    if (o->verbose)
      std::fprintf(stderr, "[sqd davidson] it %d space %d e %.12f de %.3e |r| %.3e\n", it - 1, m_res, e, de, rnorm);
    // (the same comparisons, on the same numbers, as StopRule on the device)
    if (std::fabs(de) < tol && sums[0] < tol2) {
      conv = true;
      stop = true;
    } else if (!(sums[0] > lindep) || !(sums[1] > 0.0)) {
      conv = sums[0] < tol2;
      stop = true;
    }
    return SQD_OK;
  };
  c->sigma_stop = stop_flag;
  struct StopGuard {  // the flag is only meaningful inside this run
    sqd_ctx* c;
    ~StopGuard() { c->sigma_stop = nullptr; }
  } stop_guard{c};
  for (it = 0; it < o->max_cycle; ++it) {
    // |dE| >= tol rules convergence out before the residual is known: only then is the next sigma enqueued
    // ahead of the residual hand-over (nothing is wasted except on a linear-dependence stop)
    if (have_res && std::fabs(de) < tol) {
      SQD_TRY(settle_residual());
      if (stop) break;
    }
    // sigma for the newest basis vector, and the new column of the projected matrix
    const bool timed = ev_every && nev < max_ev && (c->sigma_launches % ev_every == 0);
    ++c->sigma_launches;
    if (timed) {
      SQD_HIP_CHECK(hipEventRecord(c->sig_ev[3 * nev], s));
      c->ev_after_sigma_kernel = c->sig_ev[3 * nev + 1];  // recorded by launch_sigma right after k_sigma
    }
    const int rc_h = apply_h(c, X + (int64_t)(m - 1) * D, AX + (int64_t)(m - 1) * D, o->use_spin, o->ss, o->shift);
    c->ev_after_sigma_kernel = nullptr;
    SQD_TRY(rc_h);
    if (timed) {
      SQD_HIP_CHECK(hipEventRecord(c->sig_ev[3 * nev + 2], s));
      ++nev;
    }
    const long long seq_col = ++c->mail_seq;
    if (max_space <= 12)
      hipLaunchKernelGGL((k_dots_post<13>), dim3(gb), dim3(RED_T), 0, s, D, (const double*)X, D, m,
                         (const double*)(AX + (int64_t)(m - 1) * D), c->partial.as<double>(), width, counter, dsums_col,
                         mail_col, seq_col, (const int*)stop_flag);
    else
      hipLaunchKernelGGL((k_dots_post<SQD_MAX_SPACE + 1>), dim3(gb), dim3(RED_T), 0, s, D, (const double*)X, D, m,
                         (const double*)(AX + (int64_t)(m - 1) * D), c->partial.as<double>(), width, counter, dsums_col,
                         mail_col, seq_col, (const int*)stop_flag);
    SQD_HIP_CHECK(hipGetLastError());
    if (have_res) {
      SQD_TRY(settle_residual());
      if (stop) break;  // the sigma just enqueued is discarded
    }
    ++nsig;
    SQD_TRY(wait_mail(c, 0, seq_col, m + 1, sums.data()));
    const double nrm2 = sums[0];
    if (it == 0 && !(nrm2 > 0.0)) {
      set_error("initial vector has zero norm");
      return SQD_ERR_INVALID;
    }
    if (!(nrm2 > lindep)) {
      // the last correction vector was linearly dependent on the basis: stop with the Ritz vector of the
      // previous projected problem (pyscf: 'Linear dependency in trial subspace')
      conv = rnorm < toloose;
      --nsig;
      break;
    }
    sv.v[m - 1] = 1.0 / std::sqrt(nrm2);
    for (int i = 0; i < m; ++i)
      heff[(size_t)i * nvecs + (m - 1)] = heff[(size_t)(m - 1) * nvecs + i] = sums[1 + i] * sv.v[i] * sv.v[m - 1];
    sub.assign((size_t)m * m, 0.0);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) sub[(size_t)i * m + j] = heff[(size_t)i * nvecs + j];
    // (grown by one vector since the last projected problem: try the warm-started solver first)
    bool warm = false;
    if (use_rqi && m == m_eig + 1 && !first) {
      double e_new = 0.0;
      warm = lowest_eig_rqi(m, sub.data(), v_eig.data(), e, &e_new, v_new.data());
      if (warm) {
        w[0] = e_new;
        for (int i = 0; i < m; ++i) V[(size_t)i * m + 0] = v_new[i];
      }
    }
    if (!warm) jacobi_eigh(m, sub.data(), w.data(), V.data());
    m_eig = m;
    for (int i = 0; i < m; ++i) v_eig[i] = V[(size_t)i * m + 0];
    elast = e;
    e = w[0];
    de = first ? e : e - elast;
    first = false;
    for (int i = 0; i < m; ++i) {
      coef.v[i] = V[(size_t)i * m + 0];
      raw.v[i] = coef.v[i] * sv.v[i];  // coefficients on the stored (un-normalised) vectors
    }
    mc = m;
    // residual, preconditioned correction (into X[m]) and its overlaps; then orthogonalisation on the device
    double* tnew = X + (int64_t)m * D;
    seq_res = ++c->mail_seq;
    m_res = m;
    const StopRule rule{stop_flag, std::fabs(de) < tol ? 1 : 0, tol2, lindep};
    // (the two kernels use a partial buffer of their own: the next k_dots_post may already be enqueued)
    double* part_res = c->partial.as<double>() + (size_t)RED_BLOCKS * width;
    if (max_space <= 12) {
      hipLaunchKernelGGL((k_residual_precond<13>), dim3(gb), dim3(RED_T), 0, s, D, (const double*)X, (const double*)AX, D, m,
                         raw, e, (const double*)c->hdiag.as<double>(), pd, tnew, part_res, width);
      hipLaunchKernelGGL((k_orth_dev<13>), dim3(gb), dim3(RED_T), 0, s, D, (const double*)X, D, m, sv,
                         (const double*)part_res, (int)gb, width, tnew, mail_res, seq_res, rule, (const int*)stop_flag);
    } else {
      hipLaunchKernelGGL((k_residual_precond<SQD_MAX_SPACE + 1>), dim3(gb), dim3(RED_T), 0, s, D, (const double*)X,
                         (const double*)AX, D, m, raw, e, (const double*)c->hdiag.as<double>(), pd, tnew, part_res, width);
      hipLaunchKernelGGL((k_orth_dev<SQD_MAX_SPACE + 1>), dim3(gb), dim3(RED_T), 0, s, D, (const double*)X, D, m, sv,
                         (const double*)part_res, (int)gb, width, tnew, mail_res, seq_res, rule, (const int*)stop_flag);
    }
    SQD_HIP_CHECK(hipGetLastError());
    have_res = true;
    if (m + 1 > max_space) {
      // collapse: X0 <- Ritz vector, AX0 <- A*Ritz (linear combination), X1 <- correction
      double* x0 = c->sol.as<double>();
      SQD_TRY(c->tmp1.reserve((size_t)D * 8));
      double* ax0 = c->tmp1.as<double>();
      hipLaunchKernelGGL(k_lincomb, dim3(gb), dim3(RED_T), 0, s, D, (const double*)X, D, m, raw, x0);
      hipLaunchKernelGGL(k_lincomb, dim3(gb), dim3(RED_T), 0, s, D, (const double*)AX, D, m, raw, ax0);
      SQD_HIP_CHECK(hipMemcpyAsync(X + D, tnew, D * 8, hipMemcpyDeviceToDevice, s));
      SQD_HIP_CHECK(hipMemcpyAsync(X, x0, D * 8, hipMemcpyDeviceToDevice, s));
      SQD_HIP_CHECK(hipMemcpyAsync(AX, ax0, D * 8, hipMemcpyDeviceToDevice, s));
      std::fill(heff.begin(), heff.end(), 0.0);
      heff[0] = e;
      m = 2;
      coef.v[0] = 1.0;  // the Ritz vector is now X0
      coef.v[1] = 0.0;
      sv.v[0] = 1.0;
      mc = 1;
    } else {
      ++m;
    }
  }
  if (have_res) SQD_TRY(settle_residual());  // cycle limit reached: the last residual still decides `converged`
  // solution = Ritz vector of the last projected problem, normalised
  {
    const int mm = mc;
    double* x0 = c->sol.as<double>();
    // the basis is orthonormal and the Ritz coefficients have unit norm, so the combination is normalised to
    // rounding (the observables divide by <c|c> anyway); no extra reduction + host round trip
    double cn = 0.0;
    for (int i = 0; i < mm; ++i) cn += coef.v[i] * coef.v[i];
    Coef cf = coef;
    for (int i = 0; i < mm; ++i) cf.v[i] = (cn > 0.0 ? coef.v[i] / std::sqrt(cn) : coef.v[i]) * sv.v[i];
    hipLaunchKernelGGL(k_lincomb, dim3(gb), dim3(RED_T), 0, s, D, (const double*)X, D, mm, cf, x0);
    SQD_HIP_CHECK(hipGetLastError());
  }
  SQD_HIP_CHECK(hipEventRecord(c->ev[3], s));
  c->have_solution = true;
  c->dav_nev = nev;
  if (st) {
    st->converged = conv ? 1 : 0;
    st->iterations = it;
    st->n_sigma = nsig;
    st->e_davidson = e;
    st->residual = rnorm;
    st->ms_total = st->ms_sigma = st->ms_setup = st->ms_sigma_kernel = 0.0;
    st->n_sigma_timed = nev;
  }
  if (defer_sync) return SQD_OK;
  SQD_HIP_CHECK(hipStreamSynchronize(s));
  return davidson_collect_timings(c, st);
}

// event timings of the latest run (the stream must have been synchronised since)
int davidson_collect_timings(sqd_ctx* c, sqd_davidson_stats* st) {
  float ms = 0.f;
  SQD_HIP_CHECK(hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
  if (c->ms_setup < 0.0) {
    float tms = 0.f;
    SQD_HIP_CHECK(hipEventElapsedTime(&tms, c->ev[0], c->ev[1]));
    c->ms_setup = tms;
  }
  if (st) {
    st->ms_total = ms;
    double msig = 0.0, mker = 0.0;
    for (int i = 0; i < c->dav_nev; ++i) {
      float t = 0.f;
      SQD_HIP_CHECK(hipEventElapsedTime(&t, c->sig_ev[3 * i], c->sig_ev[3 * i + 2]));
      msig += t;
      SQD_HIP_CHECK(hipEventElapsedTime(&t, c->sig_ev[3 * i], c->sig_ev[3 * i + 1]));
      mker += t;
    }
    st->ms_sigma = msig;
    st->ms_sigma_kernel = mker;
    st->ms_setup = c->ms_setup;
  }
  return SQD_OK;
}

}  // namespace sqd
