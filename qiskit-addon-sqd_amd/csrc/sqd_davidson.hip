// Device-resident, device-CONTROLLED single-root Davidson for P (H + penalty) P.
//
// Replaces pyscf kernel_fixed_space -> FCISolver.eig -> lib.davidson1 (numpy BLAS-1 on the host)
// as called from qiskit_addon_sqd/fermion.py:721-723 and :810-818.  Control flow and constants
// follow SURVEY.md Appendix A.6: tol on |dE| and sqrt(tol) on |r|, preconditioner
// r/(hdiag - e + 1e-4), Gram-Schmidt against the whole basis, lindep drop, restart when the basis
// reaches max_space.  One deliberate difference: at a restart pyscf throws the fresh correction
// vector away and spends a sigma build on the Ritz vector; here the basis collapses to
// {Ritz vector, correction} and A*Ritz is formed by linear combination, saving that sigma build.
//
// Round-2 structure: the host no longer takes part in an iteration.  Everything an iteration decides --
// the projected matrix, its lowest eigenpair (warm-started Rayleigh-quotient iteration or Jacobi, by ONE
// wavefront inside the last workgroup of the reduction that produced the new column), the Ritz
// coefficients, the restart, the stop rule -- lives in a small state block in device memory (DavState).
// An iteration is four launches whose arguments never change,
//     k_sigma(X[m-1] -> AX[m-1])  ->  k_dots_eig  ->  k_residual_precond  ->  k_orth_dev,
// every kernel reads "which vector / how many / whether to stop" from the state block, so the host just
// keeps one iteration enqueued ahead of the one it has seen finish (a small progress record in host-
// visible memory) and no launch ever waits for the host.  Reductions are fixed-order => bitwise
// reproducible run to run.
//
// Round 6: on this chip a kernel boundary behind a pass that wrote a vector costs 4-5 us (the lines leave the L2s before
// the next kernel starts), as much as the kernels of a batch-size solve themselves.  Two boundaries went: for the
// element-gather formulation the sigma build and k_dots_eig are ONE launch (k_sigma_dots_eig: three launches per
// iteration), and the k_orth_dev launch that stops a solve forms the solution in the same pass (k_solution only runs
// when the stop came from the eigen step or the cycle limit).
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "sqd_common.h"
#include "sqd_device.h"
#include "sqd_davstate.h"
#include "sqd_direct.h"

namespace sqd {

// ---- phase clocks (probe builds only: -DSQD_PHASE_CLOCK; profiles/probes/_phase_clock.py).  Slots accumulate the
// 100 MHz wall clock over the launches since the last reset; the LAST workgroup of a launch (its critical path) writes.
#ifdef SQD_PHASE_CLOCK
__device__ unsigned long long sqd_clk[64 + 1024];  // [64 + 2 b], [65 + 2 b]: start / hand-over of workgroup b of k_dots_eig in iteration 10
#define CLK(var) const unsigned long long var = wall_clock64()
#define CLK_ACC(slot, a, b) do { if (threadIdx.x == 0) atomicAdd(&sqd_clk[slot], (unsigned long long)((b) - (a))); } while (0)
#define CLK_FIRST(slot, t) do { if (threadIdx.x == 0) atomicMin(&sqd_clk[slot], t); } while (0)
#else
#define CLK(var)
#define CLK_ACC(slot, a, b)
#define CLK_FIRST(slot, t)
#endif


constexpr int NV = 16;       // vectors per fused reduction launch (stand-alone dots)
#ifndef SQD_RED_BLOCKS_MAX
#define SQD_RED_BLOCKS_MAX 512
#endif
constexpr int RED_BLOCKS = SQD_RED_BLOCKS_MAX;
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

// Column sums of a [nblocks][width] partial-sum array by a whole workgroup, out[v] (LDS) = sum_b partial[b][v], v < nv:
// wavefront w takes the columns w, w + W, ... (W = wavefronts of the workgroup), two of them per pass; lane l adds the
// rows l, l + 64, ... (four rows in flight), a DPP tree adds the lanes.  One memory round trip, one barrier -- the
// thread-per-row form before it cost 3-4.6 us per call (a DPP tree over all nv columns in every wavefront, a cross-wave
// stage in LDS, two barriers).  Fixed order => bitwise reproducible.  Ends with a barrier: out[] may be read at once.
template <bool COHERENT>
__device__ inline void fold_partials(const double* partial, int nblocks, int width, int nv, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int v = wave; v < nv; v += 2 * nw) {
    const bool two = v + nw < nv;
    const int v2 = two ? v + nw : v;
    double s1 = 0.0, s2 = 0.0;
    for (int b0 = lane; b0 < nblocks; b0 += 256) {
      double p1[4], p2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int b = b0 + 64 * u < nblocks ? b0 + 64 * u : b0;
        const double* q = &partial[(int64_t)b * width];
        p1[u] = COHERENT ? coherent_load(q + v) : q[v];
        p2[u] = COHERENT ? coherent_load(q + v2) : q[v2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool on = b0 + 64 * u < nblocks;
        s1 += on ? p1[u] : 0.0;
        s2 += on ? p2[u] : 0.0;
      }
    }
    s1 = wave_sum_lane63(s1);
    s2 = wave_sum_lane63(s2);
    if (lane == 63) {
      out[v] = s1;
      if (two) out[v2] = s2;
    }
  }
  __syncthreads();
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

// Diagonal of the spin penalty for the preconditioner (pyscf leaves hdiag un-shifted, which makes
// the (S^2-ss)^2 form crawl; the operator is unchanged, only the preconditioner is better):
//   form 1: shift (d - ss);  form 2: shift ((d - ss)^2 + n_flip),  d = sz(sz+1) + |B\A|, n_flip = |B\A||A\B|
struct PenaltyDiag {
  int form;
  double shift, ss, szterm;
  GPtr<const uint64_t> sa;
  GPtr<const uint64_t> sb;
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

__global__ void k_dav_init(DavState* st, unsigned* counter) { dav_state_init(st, counter); }

// pyscf get_init_guess: unit vector at addr, +1e-5 on the first and -1e-5 on the last element,
// normalised here with the closed-form norm (no reduction, no host round trip)
// (every workgroup repeats the final stage of the argmin over the <= RED_BLOCKS per-block candidates
// instead of a separate single-workgroup launch)
__global__ void k_init_guess(int64_t n, const double* __restrict__ pmin, const int64_t* __restrict__ pidx, int nblocks,
                             double* __restrict__ x, DavState* st, unsigned* counter) {
  if (st && blockIdx.x == 0) dav_state_init(st, counter);
  init_guess_write(n, pmin, pidx, nblocks, x, (int64_t)blockIdx.x * blockDim.x + threadIdx.x,
                   (int64_t)gridDim.x * blockDim.x);
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

// Final stage of a stand-alone reduction + hand-over to the host in ONE single-workgroup kernel: column sums
// of the partial array (fixed order), written straight into the host-visible mailbox, then a system-scope
// fence and the sequence word.  The host spins on that word (bounded) -- no copy engine, no stream
// synchronisation on the critical path.
constexpr int MAIL_PAYLOAD = 8;  // doubles; mail[0] is the sequence word
constexpr int MAIL_SLOT = 128;   // doubles per mailbox slot: 0 stand-alone reductions, 1 Davidson progress, 2 Davidson result
template <int MAXV>
__global__ void __launch_bounds__(128) k_reduce_to_mail(const double* __restrict__ partial, int nblocks, int width, int nv,
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

// ------------------------------------------------------------------ the projected eigenproblem, on ONE wavefront
// Lane i owns row i of the small matrix (n <= MV <= 64), IN REGISTERS; a row or an element of another lane is
// fetched with v_readlane (wave_bcast with a compile-time lane), never through LDS: the first version of this
// routine kept the matrices in LDS and cost 10-25 us per iteration in dependent LDS round trips (kernel trace,
// profiles/r02), more than the host round trip it replaced.  All 64 lanes execute every collective.
//
// Lowest eigenpair after the basis grew by ONE vector, by Rayleigh-quotient iteration from the previous Ritz
// vector padded with a zero (a start whose residual is already small).  The previous matrix is the leading
// principal block of this one, so by Cauchy interlacing  l1(new) <= l1(old) <= l2(new): an eigenvalue found at or
// below the old Ritz value IS the lowest one -- that test, plus a residual at rounding level, is the acceptance
// rule; anything else returns false and the caller runs the Jacobi solver.  The shifted solves use elimination in
// the natural order (no pivot search: 2 n^2 fewer lane reads per solve).  A - theta I has at most one direction
// of the wrong sign here and the vanishing pivot is the mechanism of inverse iteration; should a solve go wrong
// all the same, the acceptance rule rejects the result and the Jacobi fallback takes over -- speed, never
// correctness, is what the missing pivoting can cost.
// a[j] = A[lane][j]; xi: previous Ritz vector component of this lane in, new one out.
// N: the order n as a compile-time constant (0: run-time n).  The loops below are unrolled to MV with a guard `j < n`
// per step; with a run-time n every guard is a scalar branch that fences the schedule (the phase clocks showed 9.5 us
// per call at a mean order of 6.5), with N the guards fold away and the body is straight-line code of exactly n steps.
template <int MV, int N>
__device__ inline bool wave_lowest_eig_rqi(int n_rt, const double (&a)[MV], double& xi_io, double e_old, double* e_out,
                                           int* n_solves) {
  const int n = N ? N : n_rt;
  const int lane = threadIdx.x & 63;
  const bool act = lane < n;
  double xi = (act && lane < n - 1) ? xi_io : 0.0;
  {
    // first-order estimate of the new vector's weight in the Ritz vector instead of a zero: -(h . v_old) / (a_nn - e_old),
    // h = coupling of the new basis vector to the old ones (row n-1, owned by lane n-1).  One dot product that usually
    // saves the second shifted solve.
    double hv = 0.0;
#pragma unroll
    for (int j = 0; j < MV; ++j)
      if (j < n - 1) hv += a[j] * wave_bcast(xi, j);
    double ann = 0.0;
#pragma unroll
    for (int j = 0; j < MV; ++j) ann = (j == n - 1) ? a[j] : ann;
    const double den = ann - e_old;
    const double delta = (fabs(den) > 1e-12) ? -hv / den : 0.0;
    const double dlast = wave_bcast(delta, n - 1);
    if (lane == n - 1 && fabs(dlast) < 0.5) xi = dlast;
  }
  double rowabs = 0.0;
#pragma unroll
  for (int j = 0; j < MV; ++j) rowabs += (j < n) ? fabs(a[j]) : 0.0;
  const double nrm = wave_sum(xi * xi), anorm = wave_max(act ? rowabs : 0.0);
  if (!(nrm > 0.0) || !(anorm > 0.0)) return false;
  xi *= 1.0 / sqrt(nrm);
  const double tiny = 2.3e-16 * anorm;
  auto matvec = [&](double xl) {  // (A x)_lane
    double r = 0.0;
#pragma unroll
    for (int j = 0; j < MV; ++j)
      if (j < n) r += a[j] * wave_bcast(xl, j);
    return act ? r : 0.0;
  };
  double yi = matvec(xi);
  double theta = wave_sum(xi * yi);
#pragma nounroll
  for (int it = 0; it < 6; ++it) {
    const double res = wave_sum(act ? (yi - theta * xi) * (yi - theta * xi) : 0.0);
    // accepted at 1e-11 |A| (the residual of the small eigenpair, not of the big one): the Ritz value is second order
    // in it and the Ritz coefficients feed a Davidson residual whose own threshold is 1e-6; rounding level (8 tiny)
    // cost a second shifted solve on nearly every call (72 solves for 35 calls on the HF-centred headline problem)
    if (sqrt(res) <= 1e-11 * anorm) {
      if (!(theta <= e_old + 64.0 * tiny)) return false;  // not provably the lowest eigenvalue
      // (a new vector that does not couple to the old Ritz vector leaves that pair an eigenpair of the grown
      // matrix although its own diagonal may lie lower: the lowest eigenvalue is below every diagonal element)
      double dg = 0.0;
#pragma unroll
      for (int j = 0; j < MV; ++j) dg = (j == lane) ? a[j] : dg;
      const double above = wave_max(act ? theta - dg : -1.0);
      if (above > 64.0 * tiny) return false;
      *e_out = theta;
      xi_io = xi;
      return true;
    }
    // z = (A - theta I)^-1 x: elimination in the natural order, row k owned by lane k
    if (lane == 0) *n_solves += 1;
    double m[MV];
#pragma unroll
    for (int j = 0; j < MV; ++j) m[j] = a[j] - ((j == lane) ? theta : 0.0);
    double z = xi;
#pragma unroll
    for (int k = 0; k < MV; ++k)
      if (k < n - 1) {
        double pk = wave_bcast(m[k], k);
        if (fabs(pk) < tiny) {
          pk = (pk < 0.0) ? -tiny : tiny;
          if (lane == k) m[k] = pk;
        }
        const double f = (act && lane > k) ? m[k] * fast_rcp(pk) : 0.0;
#pragma unroll
        for (int j = k + 1; j < MV; ++j)
          if (j < n) m[j] -= f * wave_bcast(m[j], k);
        z -= f * wave_bcast(z, k);
      }
    // back substitution: sol[k] on every lane
    double sol[MV];
#pragma unroll
    for (int k = MV - 1; k >= 0; --k) {
      sol[k] = 0.0;
      if (k < n) {
        double r = z;
#pragma unroll
        for (int j = k + 1; j < MV; ++j)
          if (j < n) r -= m[j] * sol[j];
        double pk = m[k];
        if (k == n - 1 && fabs(pk) < tiny) pk = (pk < 0.0) ? -tiny : tiny;  // the pivot that vanishes at convergence
        sol[k] = wave_bcast(r * fast_rcp(pk), k);
      }
    }
    double yl = 0.0;
#pragma unroll
    for (int k = 0; k < MV; ++k) yl = (k == lane && k < n) ? sol[k] : yl;
    const double yn = wave_sum(yl * yl), dot = wave_sum(yl * xi);
    if (!(yn > 0.0) || !(yn < 1e300)) return false;
    const double sc = ((dot < 0.0) ? -1.0 : 1.0) / sqrt(yn);  // keep the orientation of the previous Ritz vector
    xi = yl * sc;
    yi = matvec(xi);
    theta = wave_sum(xi * yi);
  }
  return false;
}

// cyclic Jacobi for a small symmetric matrix (destroyed); lowest eigenvalue returned, its eigenvector in v_out.
// Lane k owns element k of the row / column being rotated.
__device__ inline double wave_lowest_eig_jacobi(int n, int ld, double* A, double* V, double* v_out) {
  const int lane = threadIdx.x & 63;
  const bool act = lane < n;
  if (act)
    for (int j = 0; j < n; ++j) V[lane * ld + j] = (lane == j) ? 1.0 : 0.0;
  wave_sync();
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0, dia = 0.0;
    if (act) {
      dia = A[lane * ld + lane] * A[lane * ld + lane];
      for (int j = lane + 1; j < n; ++j) off += A[lane * ld + j] * A[lane * ld + j];
    }
    off = wave_sum(off);
    dia = wave_sum(dia);
    // converged to working precision (quadratic convergence: one more sweep would change nothing)
    if (off <= 1e-32 * dia || off < 1e-300) break;
    for (int p = 0; p < n; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p * ld + q], app = A[p * ld + p], aqq = A[q * ld + q];
        wave_sync();
        if (apq == 0.0) continue;  // uniform over the wave
        const double theta = (aqq - app) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
        if (act) {
          const double akp = A[lane * ld + p], akq = A[lane * ld + q];
          A[lane * ld + p] = cs * akp - sn * akq;
          A[lane * ld + q] = sn * akp + cs * akq;
        }
        wave_sync();
        if (act) {
          const double apk = A[p * ld + lane], aqk = A[q * ld + lane];
          A[p * ld + lane] = cs * apk - sn * aqk;
          A[q * ld + lane] = sn * apk + cs * aqk;
          const double vkp = V[lane * ld + p], vkq = V[lane * ld + q];
          V[lane * ld + p] = cs * vkp - sn * vkq;
          V[lane * ld + q] = sn * vkp + cs * vkq;
        }
        wave_sync();
      }
  }
  // lowest diagonal element (ties to the lower index)
  const int idx = wave_argmax(act ? -A[lane * ld + lane] : -1e300, lane);
  const double w0 = A[idx * ld + idx];
  wave_sync();
  if (act) v_out[lane] = V[lane * ld + idx];
  wave_sync();
  return w0;
}

// ---- LDS working copy of the state block.  The state lives in global memory (every kernel of an iteration reads it);
// the one wavefront that solves the projected problem used to read it field by field -- three dependent round trips, 2.6 us
// (phase clocks) -- so the workgroups of k_dots_eig now request it when they start, beside the vector loads of their main
// loop (nobody writes the block during that kernel before the last workgroup has arrived), and the eigen step works on
// the copy and stores the head back in one go.  Head = every field before `heff`; the projected matrix is kept compact
// (MV x MV) in LDS and written through.
constexpr int DAV_HEAD_WORDS = (int)(offsetof(DavState, heff) / 8);
static_assert(offsetof(DavState, heff) % 8 == 0, "the head of the state block is copied as 8-byte words");
template <int MV>
__device__ inline void dav_state_prefetch(const DavState* st, unsigned long long* s_head, double* s_heff) {
  const unsigned long long* src = reinterpret_cast<const unsigned long long*>(st);
  for (int i = threadIdx.x; i < DAV_HEAD_WORDS; i += blockDim.x) s_head[i] = src[i];
  for (int i = threadIdx.x; i < MV * MV; i += blockDim.x) s_heff[i] = st->heff[(i / MV) * MAXB + (i % MV)];
}

template <int MV, int N>
__device__ inline bool rqi_case(int m, const double (&a)[MV], double& ci, double e_old, double* e_new, int* n_solves, bool& done) {
  if (m != N) return false;
  done = wave_lowest_eig_rqi<MV, N>(m, a, ci, e_old, e_new, n_solves);
  return true;
}
// the order as a compile-time constant for the sizes of the default run (max_space 12), the run-time form beyond
template <int MV>
__device__ inline bool rqi_dispatch(int m, const double (&a)[MV], double& ci, double e_old, double* e_new, int* n_solves) {
  bool done = false;
  if (rqi_case<MV, 3>(m, a, ci, e_old, e_new, n_solves, done) || rqi_case<MV, 4>(m, a, ci, e_old, e_new, n_solves, done) ||
      rqi_case<MV, 5>(m, a, ci, e_old, e_new, n_solves, done) || rqi_case<MV, 6>(m, a, ci, e_old, e_new, n_solves, done) ||
      rqi_case<MV, 7>(m, a, ci, e_old, e_new, n_solves, done) || rqi_case<MV, 8>(m, a, ci, e_old, e_new, n_solves, done) ||
      rqi_case<MV, 9>(m, a, ci, e_old, e_new, n_solves, done) || rqi_case<MV, 10>(m, a, ci, e_old, e_new, n_solves, done) ||
      rqi_case<MV, 11>(m, a, ci, e_old, e_new, n_solves, done) || rqi_case<MV, 12>(m, a, ci, e_old, e_new, n_solves, done) ||
      rqi_case<MV, 13>(m, a, ci, e_old, e_new, n_solves, done))
    return done;
  return wave_lowest_eig_rqi<MV, 0>(m, a, ci, e_old, e_new, n_solves);
}

// One wavefront of the LAST workgroup of k_dots_eig: the new column of the projected matrix from the folded
// dot products, the linear-dependence test, the lowest eigenpair, the Ritz coefficients, the restart decision.
// tot[0] = |X_{m-1}|^2, tot[1+v] = X_v . A X_{m-1}.  w / wh: the LDS working copy (dav_state_prefetch).
template <int MV>
__device__ __attribute__((always_inline)) inline void wave_eig_step(DavState* st, unsigned long long* s_head, double* wh, const double* tot, const DavParams prm,
                                     double* sA, double* sM, double* sv_eig) {
  const int lane = threadIdx.x & 63;
  DavState* w = reinterpret_cast<DavState*>(s_head);  // (head fields only)
  CLK(e0);
  auto flush = [&]() {  // the head back to global memory: independent stores, one round trip
    wave_sync();
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(st);
    for (int i = lane; i < DAV_HEAD_WORDS; i += 64) dst[i] = s_head[i];
  };
  const int m = w->m_next;
  constexpr int LD = MV;
  if (w->restart) {  // the previous iteration collapsed the basis: X0 = Ritz vector (unit norm), A X0 by combination
    wave_sync();
    for (int i = lane; i < MV * MV; i += 64) {
      wh[i] = 0.0;
      st->heff[(i / MV) * MAXB + (i % MV)] = 0.0;
    }
    wave_sync();
    if (lane == 0) {
      wh[0] = w->e;
      st->heff[0] = w->e;
      w->sv[0] = 1.0;
      w->restart = 0;
      w->m_eig = -1;
    }
    wave_sync();
  }
  const double nrm2 = tot[0];
  if (w->it == 0 && !(nrm2 > 0.0)) {
    wave_sync();
    if (lane == 0) {
      w->err = 1;
      w->stop = 1;
    }
    flush();
    return;
  }
  if (!(nrm2 > prm.lindep)) {
    // the last correction vector was linearly dependent on the basis: stop with the Ritz vector of the
    // previous projected problem (pyscf: 'Linear dependency in trial subspace'); this sigma is not counted
    wave_sync();
    if (lane == 0) {
      w->conv = (w->rnorm2 < prm.tol2) ? 1 : 0;
      w->stop = 1;
    }
    flush();
    return;
  }
  const double svm = 1.0 / sqrt(nrm2);
  const double svi = (lane < m) ? ((lane == m - 1) ? svm : w->sv[lane]) : 0.0;
  const double e_old = w->e;
  const int m_eig = w->m_eig, first = w->first;
  // row `lane` of the projected matrix in registers: the old block from the state, the new row / column from tot
  double a[MV];
  {
    const double tl = (lane < m) ? tot[1 + lane] : 0.0;
    const int row = lane < MV ? lane : 0;
#pragma unroll
    for (int j = 0; j < MV; ++j) {
      double v = 0.0;
      if (lane < m && j < m) {
        if (lane == m - 1) v = tot[1 + j] * ((j == m - 1) ? svm : w->sv[j]) * svm;
        else if (j == m - 1) v = tl * svi * svm;
        else v = wh[row * MV + j];
      }
      a[j] = v;
    }
    if (lane < m) {
      const double hv = tl * svi * svm;
      wh[lane * MV + (m - 1)] = hv;
      wh[(m - 1) * MV + lane] = hv;
      st->heff[lane * MAXB + (m - 1)] = hv;
      st->heff[(m - 1) * MAXB + lane] = hv;
    }
  }
  double e_new = 0.0;
  double ci = 0.0;  // this lane's component of the lowest eigenvector
  bool done = false;
#ifdef SQD_PHASE_CLOCK
  __builtin_amdgcn_s_waitcnt(0);
#endif
  CLK(e1);
  if (m == 1) {
    e_new = wave_bcast(a[0], 0);
    ci = (lane == 0) ? 1.0 : 0.0;
    done = true;
  } else if (m == 2) {
    // symmetric 2 x 2 in closed form: lowest eigenvalue and its eigenvector
    const double p = wave_bcast(a[0], 0), q = wave_bcast(a[1], 0), r = wave_bcast(a[1], 1);
    const double hd = 0.5 * (p - r), rad = sqrt(hd * hd + q * q);
    e_new = 0.5 * (p + r) - rad;
    // (A - e) v = 0: v = (q, e - p) or (e - r, q); take the better conditioned one
    double v0, v1;
    if (fabs(e_new - p) > fabs(e_new - r)) {
      v0 = q;
      v1 = e_new - p;
    } else {
      v0 = e_new - r;
      v1 = q;
    }
    double nn = sqrt(v0 * v0 + v1 * v1);
    if (!(nn > 0.0)) {  // diagonal matrix with equal entries
      v0 = (p <= r) ? 1.0 : 0.0;
      v1 = 1.0 - v0;
      nn = 1.0;
    }
    const double sgn = (v0 < 0.0) ? -1.0 : 1.0;  // orientation: positive weight on the older vector
    ci = (lane == 0) ? sgn * v0 / nn : ((lane == 1) ? sgn * v1 / nn : 0.0);
    done = true;
  } else if (m == m_eig + 1 && !first) {
    ci = (lane < m - 1) ? w->coef[lane] : 0.0;  // previous Ritz vector as the warm start
    done = rqi_dispatch<MV>(m, a, ci, e_old, &e_new, &w->n_rqi);
    if (!done) ci = 0.0;
  }
  if (!done) {  // fallback: cyclic Jacobi on an LDS copy
    if (lane == 0) w->n_jacobi += 1;
    wave_sync();
    if (lane < m) {
#pragma unroll
      for (int j = 0; j < MV; ++j)
        if (j < m) sA[lane * LD + j] = a[j];
    }
    wave_sync();
    e_new = wave_lowest_eig_jacobi(m, LD, sA, sM, sv_eig);
    ci = (lane < m) ? sv_eig[lane] : 0.0;
  }
  // the Ritz coefficients of this projected problem
  CLK(e2);
  CLK_ACC(7, e0, e1);   // state + projected matrix
  CLK_ACC(8, e1, e2);   // eigenpair
  const double cn = wave_sum(ci * ci);
  wave_sync();
  if (lane < m) {
    w->coef[lane] = ci;
    w->raw[lane] = ci * svi;
    w->sol_coef[lane] = (cn > 0.0 ? ci / sqrt(cn) : ci) * svi;
    if (lane == m - 1) w->sv[lane] = svm;
  }
  if (lane == 0) {
    w->e = e_new;
    w->de = first ? e_new : e_new - e_old;
    w->first = 0;
    w->m_eig = m;
    w->m_cur = m;
    w->sol_m = m;
    w->nsig += 1;
    w->it += 1;
    const int restart = (m + 1 > prm.max_space) ? 1 : 0;
    w->restart = restart;
    w->m_next = restart ? 2 : m + 1;
  }
  flush();
}

// sums[0] = |X_{m-1}|^2, sums[1+v] = X_v . y  (v < m <= MV), y = A X_{m-1}, m = st->m_next; the workgroup that
// arrives LAST folds the per-block partials (fixed order) and one of its wavefronts solves the projected problem.
// The new sigma vector may still be in pieces: rows of C that k_sigma cut into several work items were written as
// partial rows (no atomics).  split.rowinfo != nullptr: this kernel is the first reader of that vector, so it adds
// a split row's partial rows in slot order (fixed => bitwise reproducible) and stores the finished elements -- the
// k_sigma_reduce launch that used to sit between k_sigma and here (5 us per iteration) is gone.
struct SplitRows {
  GPtr<const int32_t> rowinfo;  // [2 A] first slot, [2 A + 1] slots (0: the row was written directly)
  GPtr<const double> partial;  // [slots][nb]
  int64_t nb;
};
// (bx, nbx): this workgroup's index in ITS subspace's grid and that grid's size -- the whole launch for the single
// kernels, one z-slice for the batched ones (sqd_solve_batch)
// FUSED: the workgroup that arrives last goes on to fold the partials and solve the projected problem (single solves:
// one launch less on a latency-bound chain).  !FUSED: the dot products only -- the eigen step's 232 VGPRs would hold this
// kernel to one workgroup per CU, and a batched launch is 16 x 197 workgroups of bandwidth-bound work; k_eig_b follows.
// Either way the partials, their fold (fold_partials in a RED_T-thread workgroup) and the eigen step are the same code:
// the same bits.
// (TOTALS -- row-sharded solves: the workgroup that arrives last folds the partials into tot_out, which the caller
// all-reduces over the ranks before k_shard_eig; FUSED must be false)
// The element loop of the dot-product kernels with the basis slots bounded by K >= nvec instead of MV.  Slots past nvec
// are loaded all the same (load_vectors: branch-free, an L1 hit) and multiplied by nothing -- free while a launch is a
// chain of round trips (D = 1e5: one element per thread), but at D = 1e8 a thread walks ~400 elements and thirteen
// loads and FMA pairs per element for a basis of two were most of the kernel: 0.81 ms for 1.6 GB at m = 1, growing by
// only 0.06 ms per vector (profiles/r04b/big_davidson_probe.txt).  The arithmetic per element is the same for every K.
template <int MV, int K>
__device__ inline void dots_loop(int64_t n, const double* __restrict__ X, double* __restrict__ y, int64_t stride, int nvec,
                                 const SplitRows& split, unsigned bx, unsigned nbx, double (&acc)[MV + 1]) {
  static_assert(K <= MV, "slot bound");
  for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < n; i += (int64_t)nbx * blockDim.x) {
    // every request that does not depend on another one first: the basis vectors, y, the row's split record -- the
    // partial rows of a split row are the only second round trip (they were the third)
    double xv[K];
    load_vectors<K>(X, stride, nvec, i, xv);
    double yv = y[i];
    if (split.rowinfo) {
      const int64_t A = (n < (int64_t)1 << 31) ? (int64_t)((unsigned)i / (unsigned)split.nb) : i / split.nb;
      const int64_t B = i - A * split.nb;
      const int2 info = *reinterpret_cast<const int2*>(&split.rowinfo[2 * A]);
      const int slot0 = info.x, ns = info.y;
      if (ns > 0) {
        double sacc = 0.0;
        for (int j0 = 0; j0 < ns; j0 += 16) {  // eight or sixteen partial rows in flight per round, added in slot order
          double pv[16];
#pragma unroll
          for (int u = 0; u < 8; ++u) pv[u] = split.partial[(int64_t)(slot0 + (j0 + u < ns ? j0 + u : j0)) * split.nb + B];
          if (j0 + 8 < ns) {
#pragma unroll
            for (int u = 8; u < 16; ++u) pv[u] = split.partial[(int64_t)(slot0 + (j0 + u < ns ? j0 + u : j0)) * split.nb + B];
          } else {
#pragma unroll
            for (int u = 8; u < 16; ++u) pv[u] = 0.0;
          }
#pragma unroll
          for (int u = 0; u < 16; ++u) sacc += (j0 + u < ns) ? pv[u] : 0.0;
        }
        yv = sacc;
        y[i] = sacc;
      }
    }
#pragma unroll
    for (int v = 0; v < K; ++v) {
      acc[1 + v] += (v < nvec) ? xv[v] * yv : 0.0;
      acc[0] += (v == nvec - 1) ? xv[v] * xv[v] : 0.0;
    }
  }
}
template <int MV>
__device__ inline void dots_loop_select(int64_t n, const double* __restrict__ X, double* __restrict__ y, int64_t stride, int nvec,
                                        const SplitRows& split, unsigned bx, unsigned nbx, double (&acc)[MV + 1]) {
  if (nvec <= 2) dots_loop<MV, (2 < MV ? 2 : MV)>(n, X, y, stride, nvec, split, bx, nbx, acc);
  else if (nvec <= 4) dots_loop<MV, (4 < MV ? 4 : MV)>(n, X, y, stride, nvec, split, bx, nbx, acc);
  else if (nvec <= 8) dots_loop<MV, (8 < MV ? 8 : MV)>(n, X, y, stride, nvec, split, bx, nbx, acc);
  else dots_loop<MV, MV>(n, X, y, stride, nvec, split, bx, nbx, acc);
}
template <int MV, bool FUSED, bool TOTALS = false>
__device__ inline void dots_eig_body(int64_t n, const double* __restrict__ X, double* __restrict__ AX, int64_t stride,
                                     double* __restrict__ partial, int width, unsigned* counter, DavState* st,
                                     const DavParams& prm, const SplitRows& split, unsigned bx, unsigned nbx,
                                     double* __restrict__ tot_out = nullptr) {
  __shared__ double red[16 * (MV + 1)];
  __shared__ double tot[FUSED ? MV + 1 : 1];
  __shared__ double sA[FUSED ? MV * MV : 1], sM[FUSED ? MV * MV : 1], sv_eig[FUSED ? MV + 1 : 1];
  __shared__ unsigned long long s_head[FUSED ? DAV_HEAD_WORDS : 1];
  __shared__ double s_heff[FUSED ? MV * MV : 1];
  CLK(c0);
  CLK_FIRST(63, c0);
  if (st->stop) return;  // enqueued behind the iteration that ended the solve (nobody writes the flag during this
                         // kernel before every workgroup has arrived)
  const int nvec = st->m_next;
  if (FUSED) dav_state_prefetch<MV>(st, s_head, s_heff);  // (for the workgroup that turns out to be the last one)
  double* __restrict__ y = AX + (int64_t)(nvec - 1) * stride;
  double acc[MV + 1];
#pragma unroll
  for (int v = 0; v < MV + 1; ++v) acc[v] = 0.0;
  dots_loop_select<MV>(n, X, y, stride, nvec, split, bx, nbx, acc);
  block_sum_multi<MV + 1>(acc, nvec + 1, red);
  if constexpr (!FUSED && !TOTALS) {
    if ((int)threadIdx.x < nvec + 1) partial[(int64_t)bx * width + threadIdx.x] = block_sum_multi_get<MV + 1>(red, threadIdx.x);
    return;
  }
  if constexpr (TOTALS) {
    __shared__ double s_tot[MV + 1];
    if ((int)threadIdx.x < nvec + 1)
      coherent_store(&partial[(int64_t)bx * width + threadIdx.x], block_sum_multi_get<MV + 1>(red, threadIdx.x));
    if (!arrive_last(counter, bx, nbx)) return;
    fold_partials<true>(partial, (int)nbx, width, nvec + 1, s_tot);
    if ((int)threadIdx.x <= MAXB) tot_out[threadIdx.x] = ((int)threadIdx.x < nvec + 1) ? s_tot[threadIdx.x] : 0.0;
    return;
  }
  if ((int)threadIdx.x < nvec + 1)
    coherent_store(&partial[(int64_t)bx * width + threadIdx.x], block_sum_multi_get<MV + 1>(red, threadIdx.x));
  CLK(c1);
#ifdef SQD_PHASE_CLOCK
  if (threadIdx.x == 0 && st->it == 10 && bx < 512) {
    sqd_clk[64 + 2 * bx] = c0;
    sqd_clk[65 + 2 * bx] = c1;
  }
#endif
  if (!arrive_last(counter, bx, nbx)) return;
  CLK(c2);
  fold_partials<true>(partial, (int)nbx, width, nvec + 1, tot);
  if (threadIdx.x >= 64) return;
  CLK(c3);
  wave_eig_step<MV>(st, s_head, s_heff, tot, prm, sA, sM, sv_eig);
#ifdef SQD_PHASE_CLOCK
  __builtin_amdgcn_s_waitcnt(0);
  CLK(c4);
  if (threadIdx.x == 0) {
    const unsigned long long first = atomicExch(&sqd_clk[63], ~0ull);
    atomicAdd(&sqd_clk[0], 1ull);          // launches
    atomicAdd(&sqd_clk[1], c0 - first);    // first workgroup's start -> last workgroup's start
    atomicAdd(&sqd_clk[2], c1 - c0);       // main loop + block sum + partial store (last workgroup)
    atomicAdd(&sqd_clk[3], c2 - c1);       // arrival
    atomicAdd(&sqd_clk[4], c3 - c2);       // fold
    atomicAdd(&sqd_clk[5], c4 - c3);       // eig step
    atomicAdd(&sqd_clk[6], (unsigned long long)st->m_cur);
  }
#endif
}
template <int MV>
__global__ void __launch_bounds__(RED_T) k_dots_eig(int64_t n, const double* __restrict__ X, double* __restrict__ AX, int64_t stride,
                           double* __restrict__ partial, int width, unsigned* counter, DavState* st,
                           const DavParams prm, const SplitRows split) {
  dots_eig_body<MV, true>(n, X, AX, stride, partial, width, counter, st, prm, split, blockIdx.x, gridDim.x);
}

// ---- The sigma build AND the fused reduction behind it in ONE launch, for the element-gather formulation (round 6).  That
// sigma kernel is a thread per output element with no exchange between threads, and the dot products need sigma[i] only
// where they need X_v[i]: the thread that gathers element i multiplies it into its partial sums while the value is in a
// register.  One kernel boundary less per iteration on a chain whose kernels run 5-8 us and whose boundaries cost 4-5 (a
// pass that wrote a vector has to drain the L2s before the next kernel starts): uniform 317 x 317, 2-3 iterations per
// solve.  Geometry, loop order and reductions are k_dots_eig's, the element is k_sigma_direct's (direct_element): the
// same bits as the two launches, which the batched solves and the timing hooks keep (tests compare them).
template <int MV, int K, bool SPIN>
__device__ inline void dots_loop_direct(int64_t n, const double* __restrict__ X, double* __restrict__ y, int64_t stride, int nvec,
                                        const DirectArgs& dg, unsigned bx, unsigned nbx, double (&acc)[MV + 1]) {
  static_assert(K <= MV, "slot bound");
  const double* __restrict__ Cn = X + (int64_t)(nvec - 1) * stride;  // the newest basis vector: sigma's operand
  const double pen = -dg.shift;
  for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < n; i += (int64_t)nbx * blockDim.x) {
    double xv[K];
    load_vectors<K>(X, stride, nvec, i, xv);
    const double yv = direct_element<SPIN>(dg, Cn, i, pen);
    y[i] = yv;
#pragma unroll
    for (int v = 0; v < K; ++v) {
      acc[1 + v] += (v < nvec) ? xv[v] * yv : 0.0;
      acc[0] += (v == nvec - 1) ? xv[v] * xv[v] : 0.0;
    }
  }
}
template <int MV, bool SPIN>
__global__ void __launch_bounds__(RED_T) k_sigma_dots_eig(int64_t n, const double* __restrict__ X, double* __restrict__ AX,
                                                           int64_t stride, double* __restrict__ partial, int width,
                                                           unsigned* counter, DavState* st, const DavParams prm,
                                                           const DirectArgs dg) {
  __shared__ double red[16 * (MV + 1)];
  __shared__ double tot[MV + 1];
  __shared__ double sA[MV * MV], sM[MV * MV], sv_eig[MV + 1];
  __shared__ unsigned long long s_head[DAV_HEAD_WORDS];
  __shared__ double s_heff[MV * MV];
  if (st->stop) return;
  const unsigned bx = blockIdx.x, nbx = gridDim.x;
  const int nvec = st->m_next;
  dav_state_prefetch<MV>(st, s_head, s_heff);
  double* __restrict__ y = AX + (int64_t)(nvec - 1) * stride;
  double acc[MV + 1];
#pragma unroll
  for (int v = 0; v < MV + 1; ++v) acc[v] = 0.0;
  if (nvec <= 2) dots_loop_direct<MV, (2 < MV ? 2 : MV), SPIN>(n, X, y, stride, nvec, dg, bx, nbx, acc);
  else if (nvec <= 4) dots_loop_direct<MV, (4 < MV ? 4 : MV), SPIN>(n, X, y, stride, nvec, dg, bx, nbx, acc);
  else if (nvec <= 8) dots_loop_direct<MV, (8 < MV ? 8 : MV), SPIN>(n, X, y, stride, nvec, dg, bx, nbx, acc);
  else dots_loop_direct<MV, MV, SPIN>(n, X, y, stride, nvec, dg, bx, nbx, acc);
  block_sum_multi<MV + 1>(acc, nvec + 1, red);
  if ((int)threadIdx.x < nvec + 1)
    coherent_store(&partial[(int64_t)bx * width + threadIdx.x], block_sum_multi_get<MV + 1>(red, threadIdx.x));
  if (!arrive_last(counter, bx, nbx)) return;
  fold_partials<true>(partial, (int)nbx, width, nvec + 1, tot);
  if (threadIdx.x >= 64) return;
  wave_eig_step<MV>(st, s_head, s_heff, tot, prm, sA, sM, sv_eig);
}

// Single solves of large subspaces (D >= DOTS_SPLIT_D): dots and eigen step as two launches, as the batched solves run
// them -- the eigen step's code holds the fused kernel at 232 VGPRs, i.e. ONE 512-thread workgroup per CU, and from
// D ~ 5e5 on the launch is 512 workgroups of bandwidth-bound work (HF-centred 1000^2: 44 us of a 290 us iteration);
// alone the dot products need 107 VGPRs.  The fold and the eigen step are the same code on the same partials: the same
// bits as the fused kernel (tested).  Below that size the fused kernel saves a dispatch on a latency-bound chain.
template <int MV>
__global__ void __launch_bounds__(RED_T, 4) k_dots_s(int64_t n, const double* __restrict__ X, double* __restrict__ AX, int64_t stride,
                                                      double* __restrict__ partial, int width, unsigned* counter, DavState* st,
                                                      const DavParams prm, const SplitRows split) {
  dots_eig_body<MV, false>(n, X, AX, stride, partial, width, counter, st, prm, split, blockIdx.x, gridDim.x);
}
template <int MV>
__global__ void __launch_bounds__(RED_T) k_eig_s(const double* __restrict__ partial, int gb, int width, DavState* st,
                                                  const DavParams prm) {
  __shared__ double tot[MV + 1];
  __shared__ double sA[MV * MV], sM[MV * MV], sv_eig[MV + 1];
  __shared__ unsigned long long s_head[DAV_HEAD_WORDS];
  __shared__ double s_heff[MV * MV];
  if (st->stop) return;
  const int nvec = st->m_next;
  dav_state_prefetch<MV>(st, s_head, s_heff);
  fold_partials<false>(partial, gb, width, nvec + 1, tot);
  if (threadIdx.x >= 64) return;
  wave_eig_step<MV>(st, s_head, s_heff, tot, prm, sA, sM, sv_eig);
}

// r = sum_v raw[v] (AX_v - e X_v);  t = r / (hdiag - e + 1e-4);  t stored to X[m].
// partial[block*width + {0: |r|^2, 1: |t|^2, 2+v: X_v . t}]; k_orth_dev (next in the stream) folds them.
// FOLD (row-sharded solves): the workgroup that arrives last folds the partials into tot_out (what the all-reduce over
// the ranks reads) -- the single solver leaves the fold to every workgroup of k_orth_dev, which a collective in between
// rules out, and a one-workgroup fold launch cost a dispatch and 3-4 us per iteration
// (the element loop with the basis slots bounded by K >= nvec: see dots_loop)
template <int MV, int K>
__device__ inline void residual_loop(int64_t n, double* __restrict__ X, const double* __restrict__ AX, int64_t stride, int nvec,
                                     double e, const double* __restrict__ hdiag, const PenaltyDiag& pd, const double* s_raw,
                                     double* __restrict__ out, unsigned bx, unsigned nbx, double (&vals)[MV + 2]) {
  static_assert(K <= MV, "slot bound");
  for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < n; i += (int64_t)nbx * blockDim.x) {
    double r = 0.0;
    double xv[K];
    const double hd = hdiag[i];
    // eight (X_v, AX_v) pairs requested per round, branch-free (see load_vectors); X_v is kept for the overlaps
#pragma unroll
    for (int v0 = 0; v0 < K; v0 += 8) {
      double a8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v0 + u < K) {
          const int64_t off = (int64_t)(v0 + u < nvec ? v0 + u : 0) * stride + i;
          xv[v0 + u] = X[off];
          a8[u] = AX[off];
        }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v0 + u < K) r += (v0 + u < nvec) ? s_raw[v0 + u] * (a8[u] - e * xv[v0 + u]) : 0.0;
    }
    const double t = r / (hd + penalty_diag(pd, i) - e + 1e-4);
    out[i] = t;
    vals[0] += r * r;
    vals[1] += t * t;
#pragma unroll
    for (int v = 0; v < K; ++v) vals[2 + v] += (v < nvec) ? xv[v] * t : 0.0;
  }
}
template <int MV, bool FOLD = false>
__device__ inline void residual_precond_body(int64_t n, double* __restrict__ X, const double* __restrict__ AX, int64_t stride,
                                             const DavState* __restrict__ st, const double* __restrict__ hdiag,
                                             const PenaltyDiag& pd, double* __restrict__ partial, int width, unsigned bx,
                                             unsigned nbx, unsigned* counter = nullptr, double* __restrict__ tot_out = nullptr) {
  // vals[0] = |r|^2, vals[1] = |t|^2, vals[2+v] = X_v . t ; MV bounds the basis size (registers)
  __shared__ double red[16 * (MV + 2)];
  __shared__ double s_raw[MV];
  CLK(r0);
  if (st->stop) return;
  const int nvec = st->m_cur;
  const double e = st->e;
  if ((int)threadIdx.x < MV) s_raw[threadIdx.x] = ((int)threadIdx.x < nvec) ? st->raw[threadIdx.x] : 0.0;
  __syncthreads();
  CLK(r1);
  double* __restrict__ out = X + (int64_t)nvec * stride;
  double vals[MV + 2];
#pragma unroll
  for (int v = 0; v < MV + 2; ++v) vals[v] = 0.0;
  if (nvec <= 2) residual_loop<MV, (2 < MV ? 2 : MV)>(n, X, AX, stride, nvec, e, hdiag, pd, s_raw, out, bx, nbx, vals);
  else if (nvec <= 4) residual_loop<MV, (4 < MV ? 4 : MV)>(n, X, AX, stride, nvec, e, hdiag, pd, s_raw, out, bx, nbx, vals);
  else if (nvec <= 8) residual_loop<MV, (8 < MV ? 8 : MV)>(n, X, AX, stride, nvec, e, hdiag, pd, s_raw, out, bx, nbx, vals);
  else residual_loop<MV, MV>(n, X, AX, stride, nvec, e, hdiag, pd, s_raw, out, bx, nbx, vals);
  CLK(r2);
  block_sum_multi<MV + 2>(vals, nvec + 2, red);
  if constexpr (FOLD) {
    __shared__ double s_tot[MV + 2];
    if ((int)threadIdx.x < nvec + 2)
      coherent_store(&partial[(int64_t)bx * width + threadIdx.x], block_sum_multi_get<MV + 2>(red, threadIdx.x));
    if (!arrive_last(counter, bx, nbx)) return;
    fold_partials<true>(partial, (int)nbx, width, nvec + 2, s_tot);
    if ((int)threadIdx.x <= MAXB + 1) tot_out[threadIdx.x] = ((int)threadIdx.x < nvec + 2) ? s_tot[threadIdx.x] : 0.0;
    return;
  }
  if ((int)threadIdx.x < nvec + 2)
    partial[(int64_t)bx * width + threadIdx.x] = block_sum_multi_get<MV + 2>(red, threadIdx.x);
#ifdef SQD_PHASE_CLOCK
  __builtin_amdgcn_s_waitcnt(0);
  CLK(r3);
  if (bx == 0 && threadIdx.x == 0) {
    atomicAdd(&sqd_clk[10], 1ull);
    atomicAdd(&sqd_clk[11], r1 - r0);  // state loads
    atomicAdd(&sqd_clk[12], r2 - r1);  // main loop
    atomicAdd(&sqd_clk[13], r3 - r2);  // block sum + store
  }
#endif
}
#ifndef SQD_RESID_WAVES
#define SQD_RESID_WAVES 1
#endif
template <int MV>
__global__ void __launch_bounds__(RED_T, SQD_RESID_WAVES) k_residual_precond(int64_t n, double* __restrict__ X, const double* __restrict__ AX, int64_t stride,
                                   const DavState* __restrict__ st, const double* __restrict__ hdiag,
                                   const PenaltyDiag pd, double* __restrict__ partial, int width) {
  residual_precond_body<MV>(n, X, AX, stride, st, hdiag, pd, partial, width, blockIdx.x, gridDim.x);
}

// progress record of one iteration in host-visible memory: {sequence word | it, stop, e, de, |r|^2, m}
__device__ inline void post_progress(double* mail, long long seq, const DavState* st, int stop, double rr, int sol_done = 0) {
  mail_store(&mail[MAIL_PAYLOAD + 6], (double)sol_done);
  mail_store(&mail[MAIL_PAYLOAD + 0], (double)st->it);
  mail_store(&mail[MAIL_PAYLOAD + 1], (double)stop);
  mail_store(&mail[MAIL_PAYLOAD + 2], st->e);
  mail_store(&mail[MAIL_PAYLOAD + 3], st->de);
  mail_store(&mail[MAIL_PAYLOAD + 4], rr);
  mail_store(&mail[MAIL_PAYLOAD + 5], (double)st->m_cur);
  __builtin_amdgcn_s_waitcnt(0);
  __threadfence_system();
  *reinterpret_cast<volatile long long*>(mail) = seq;
}

// t <- scale * t - sum_v g_v X_v with everything derived on the device from the residual kernel's totals
// (tot = {|r|^2, |t|^2, X_v . t}) and the per-vector normalisation factors sv (basis vector v is
// sv_v * X_v): g'_v = sv_v (X_v . t) / |t|, c2 = sum g'_v^2.  The basis is orthonormal, so
// |t/|t| - sum g'_v sv_v X_v|^2 = 1 - c2 is known before the vector is formed: when 1 - c2 > 1e-3 the
// result is normalised in the same pass; otherwise it is left with its true (small) norm.  Either way the
// next k_dots_eig measures |X_new|^2 and carries 1/sqrt of it as sv_new, so no separate normalisation pass.
//
// The residual kernel leaves only per-workgroup partials: EVERY workgroup here folds them itself (fixed
// order, the same arithmetic everywhere => the same totals to the bit), which is cheaper than a finishing
// step inside the residual kernel (arrival atomics + coherent re-read, ~8 us) and needs no extra launch.
// Every workgroup applies the stop rule to itself; workgroup 0 records it in the state block (for the kernels
// enqueued behind) and posts the iteration's progress record to the host.
// When the iteration is a restart (basis full), the same pass collapses the basis: X0 <- Ritz vector,
// AX0 <- A * Ritz (linear combinations, element by element in place), X1 <- the correction.
template <int G>
__device__ inline void orth_loop(int64_t n, const double* __restrict__ X, int64_t stride, int nvec, double scale, const double* g,
                                 double* __restrict__ t, double* __restrict__ send, unsigned bx, unsigned nbx) {
  for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < n; i += (int64_t)nbx * blockDim.x) {
    double s = scale * t[i];
    for (int v0 = 0; v0 < nvec; v0 += G) {  // G vectors' loads in flight per round
      double x[G];
#pragma unroll
      for (int u = 0; u < G; ++u) x[u] = X[(int64_t)(v0 + u < nvec ? v0 + u : v0) * stride + i];
#pragma unroll
      for (int u = 0; u < G; ++u) s -= (v0 + u < nvec) ? g[v0 + u < nvec ? v0 + u : v0] * x[u] : 0.0;
    }
    t[i] = s;
    if (send) send[i] = s;
  }
}
template <int MV>
__device__ inline void orth_dev_body(int64_t n, double* __restrict__ X, double* __restrict__ AX, int64_t stride, DavState* st,
                                     const DavParams& prm, const double* __restrict__ partial, int nblocks, int width,
                                     double* mail, long long seq, unsigned bx, unsigned nbx,
                                     const double* __restrict__ tot_in = nullptr, double* __restrict__ send = nullptr,
                                     double* __restrict__ sol_out = nullptr, double* res_out = nullptr) {
  // sol_out != nullptr (single solves): the launch that stops the solve forms the solution in the same pass -- the Ritz
  // vector of the projected problem this iteration solved -- and leaves the run's outcome in res_out, as k_solution would
  // one dispatch later (a kernel boundary costs 4-5 us on this chain; a solve of the headline is 2-3 iterations long)
  // send != nullptr (row-sharded solves): the new vector is also written to the buffer the next iteration's all-gather
  // reads -- the k_shard_pick launch that copied it there is needed for the start vector only
  // tot_in != nullptr (row-sharded solves): the totals {|r|^2, |t|^2, X_v . t} are already folded AND all-reduced
  // over the ranks; every workgroup of every rank reads the same numbers and takes the same decisions
  __shared__ double red[16 * (MV + 2)];
  __shared__ double tot[MV + 2];
  __shared__ double g[MV + 2];
  __shared__ double s_raw[MV];
  __shared__ double s_scale;
  __shared__ int s_stop, s_was_stopped;
  // (workgroup 0 raises st->stop further down while other workgroups may still be starting: the flag is
  // sampled once per workgroup so that all its threads take the same path)
  CLK(o0);
  const int my_mark = (int)(seq & 0x3fffffff) + 2;  // (>= 2: DavState::stop)
  if (threadIdx.x == 0) {
    // (a flag raised by THIS launch carries its mark: this workgroup then goes on, reaches the same stop decision from the
    // same totals and forms its share of the solution)
    const int up = st->stop;
    s_was_stopped = (up != 0 && up != my_mark) ? 1 : 0;
  }
  __syncthreads();
  if (s_was_stopped) {
    if (bx == 0 && threadIdx.x == 0) post_progress(mail, seq, st, 1, st->rnorm2, (sol_out && st->stop >= 2) ? 1 : 0);
    return;
  }
  const int nvec = st->m_cur;
  const int restart = st->restart;
  const int nv = nvec + 2;
  if (tot_in) {
    if ((int)threadIdx.x < nv) tot[threadIdx.x] = tot_in[threadIdx.x];
  } else {
    fold_partials<false>(partial, nblocks, width, nv, tot);
  }
  __syncthreads();
  CLK(o1);
  const double rr = tot[0], tt = tot[1];
  // one wavefront, lane v = basis vector v: its state words are requested together (thread 0 alone used to walk the
  // basis twice with a global load per step: 4.6 us of every workgroup's life, phase clocks of round 3)
  if (threadIdx.x < 64) {
    const int v = threadIdx.x;
    const bool on = v < nvec && v < MV;
    const double svv = on ? st->sv[v] : 0.0, rawv = on ? st->raw[v] : 0.0;
    const double de = st->de;
    const double gv = (on && tt > 0.0) ? svv * tot[2 + v] / sqrt(tt) : 0.0;
    const double c2 = wave_sum(gv * gv);
    const double inv = (1.0 - c2 > 1e-3) ? 1.0 / sqrt(1.0 - c2) : 1.0;
    if (v < MV) {
      g[v] = gv * inv * svv;
      s_raw[v] = rawv;
    }
    if (v == 0) {
      s_scale = (tt > 0.0) ? inv / sqrt(tt) : 0.0;
      const int de_small = fabs(de) < prm.tol;
      s_stop = ((de_small && rr < prm.tol2) || !(rr > prm.lindep) || !(tt > 0.0)) ? 1 : 0;
      if (bx == 0) {
        st->rnorm2 = rr;
        if (s_stop) {
          st->conv = (rr < prm.tol2) ? 1 : 0;
          st->stop = my_mark;
        } else if (restart) {
          // from here on the solution is X0 alone (the collapse below), should the run end before the next
          // projected problem is solved
          st->sol_coef[0] = 1.0;
          st->sol_m = 1;
        }
      }
    }
  }
  __syncthreads();
  CLK(o2);
  // the iteration's progress record for the host: by the LAST wavefront of workgroup 0, beside the others' main loop
  if (bx == 0 && threadIdx.x == blockDim.x - 64) post_progress(mail, seq, st, s_stop, rr, (s_stop && sol_out) ? 1 : 0);
  if (s_stop) {  // the correction is not needed (and may be 0/0)
    if (sol_out) {
      // (workgroup 0's thread 0 has just written conv / stop: it is the thread that writes the outcome record)
      if (bx == 0 && threadIdx.x == 0) {
        mail_store(&res_out[0], (double)((rr < prm.tol2) ? 1 : 0));
        mail_store(&res_out[1], (double)st->it);
        mail_store(&res_out[2], (double)st->nsig);
        mail_store(&res_out[3], st->e);
        mail_store(&res_out[4], rr);
        mail_store(&res_out[5], (double)st->err);
        mail_store(&res_out[6], 1.0);
        mail_store(&res_out[7], (double)st->n_rqi);
        mail_store(&res_out[8], (double)st->n_jacobi);
      }
      // sum_{v < sol_m} sol_coef[v] X_v, as solution_body forms it (sol_m = nvec: this iteration's projected problem)
      if ((int)threadIdx.x < MV + 2) g[threadIdx.x] = ((int)threadIdx.x < nvec) ? st->sol_coef[threadIdx.x] : 0.0;
      __syncthreads();
      for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < n; i += (int64_t)nbx * blockDim.x) {
        double sacc = 0.0;
        for (int v0 = 0; v0 < nvec; v0 += 8) {
          double x[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) x[u] = X[(int64_t)(v0 + u < nvec ? v0 + u : v0) * stride + i];
#pragma unroll
          for (int u = 0; u < 8; ++u) sacc += (v0 + u < nvec) ? g[v0 + u < nvec ? v0 + u : v0] * x[u] : 0.0;
        }
        sol_out[i] = sacc;
      }
    }
    return;
  }
  const double scale = s_scale;
  double* __restrict__ t = X + (int64_t)nvec * stride;
  if (!restart) {
    // (rounds of G vectors' loads in flight; G = 2 / 4 for small bases: the slots past nvec of a round are loaded and
    // dropped -- see dots_loop)
    if (nvec <= 2) orth_loop<2>(n, X, stride, nvec, scale, g, t, send, bx, nbx);
    else if (nvec <= 4) orth_loop<4>(n, X, stride, nvec, scale, g, t, send, bx, nbx);
    else orth_loop<8>(n, X, stride, nvec, scale, g, t, send, bx, nbx);
#ifdef SQD_PHASE_CLOCK
    __builtin_amdgcn_s_waitcnt(0);
    CLK(o3);
    if (bx == 0 && threadIdx.x == 0) {
      atomicAdd(&sqd_clk[20], 1ull);
      atomicAdd(&sqd_clk[21], o1 - o0);  // stop flag + fold of the residual kernel's partials
      atomicAdd(&sqd_clk[22], o2 - o1);  // coefficients, stop rule
      atomicAdd(&sqd_clk[23], o3 - o2);  // main loop
    }
    if (bx == 0 && threadIdx.x == blockDim.x - 64) atomicAdd(&sqd_clk[24], o3 - o2);  // progress record + main loop (the posting wavefront)
#endif
    return;
  }
  // restart: the same pass also forms the Ritz vector and A * Ritz, in place, element by element
  for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < n; i += (int64_t)nbx * blockDim.x) {
    double s = scale * t[i];
    double x0 = 0.0, ax0 = 0.0;
    for (int v0 = 0; v0 < nvec; v0 += 8) {
      double x[8], a[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t off = (int64_t)(v0 + u < nvec ? v0 + u : v0) * stride + i;
        x[u] = X[off];
        a[u] = AX[off];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int v = v0 + u < nvec ? v0 + u : v0;
        const double on = (v0 + u < nvec) ? 1.0 : 0.0;
        s -= on * g[v] * x[u];
        x0 += on * s_raw[v] * x[u];
        ax0 += on * s_raw[v] * a[u];
      }
    }
    X[i] = x0;
    AX[i] = ax0;
    X[stride + i] = s;
    if (send) send[i] = s;
  }
}
template <int MV>
__global__ void __launch_bounds__(RED_T) k_orth_dev(int64_t n, double* __restrict__ X, double* __restrict__ AX, int64_t stride, DavState* st,
                           const DavParams prm, const double* __restrict__ partial, int nblocks, int width,
                           double* mail, long long seq, double* __restrict__ sol_out, double* res_out) {
  orth_dev_body<MV>(n, X, AX, stride, st, prm, partial, nblocks, width, mail, seq, blockIdx.x, gridDim.x, nullptr, nullptr,
                    sol_out, res_out);
}

// the solution: sum_{v < sol_m} sol_coef[v] X_v (unit norm: orthonormal basis, unit Ritz coefficients), and the
// run's outcome for the host (read after the stream has been synchronised)
__device__ inline void solution_body(int64_t n, const double* __restrict__ X, int64_t stride, const DavState* __restrict__ st,
                                     double* __restrict__ out, double* res, unsigned bx, unsigned nbx) {
  __shared__ double s_c[MAXB + 1];
  const int nvec = st->sol_m;
  if ((int)threadIdx.x <= MAXB) s_c[threadIdx.x] = ((int)threadIdx.x < nvec) ? st->sol_coef[threadIdx.x] : 0.0;
  __syncthreads();
  if (bx == 0 && threadIdx.x == 0) {
    mail_store(&res[0], (double)st->conv);
    mail_store(&res[1], (double)st->it);
    mail_store(&res[2], (double)st->nsig);
    mail_store(&res[3], st->e);
    mail_store(&res[4], st->rnorm2);
    mail_store(&res[5], (double)st->err);
    mail_store(&res[6], (double)(st->stop != 0));
    mail_store(&res[7], (double)st->n_rqi);
    mail_store(&res[8], (double)st->n_jacobi);
  }
  for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < n; i += (int64_t)nbx * blockDim.x) {
    double s = 0.0;
    for (int v0 = 0; v0 < nvec; v0 += 8) {
      double x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = X[(int64_t)(v0 + u < nvec ? v0 + u : v0) * stride + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += (v0 + u < nvec) ? s_c[v0 + u < nvec ? v0 + u : v0] * x[u] : 0.0;
    }
    out[i] = s;
  }
}
// only_if_stopped: a launch queued speculatively behind a round of the first few (run_davidson): it forms the solution
// if that round stopped the solve and is an early exit otherwise
__global__ void k_solution(int64_t n, const double* __restrict__ X, int64_t stride, const DavState* __restrict__ st,
                           double* __restrict__ out, double* res, int only_if_stopped) {
  if (only_if_stopped && !st->stop) return;
  solution_body(n, X, stride, st, out, res, blockIdx.x, gridDim.x);
}

// ---- row-sharded solves (reference docs/guides/hpc_acceleration.rst:52-57: a collective sci_solver).  Every
// rank holds the rows [row0, row1) of all Davidson vectors and the SAME state block: the reductions are cut where a
// number has to cross ranks -- local totals to a device buffer, an all-reduce by the caller on the same stream, the rest
// of the iteration from the reduced totals -- so that the state machine (projected matrix, eigenpair, restart, stop
// rule) is this file's, device-resident, and evolves bit-identically on every rank.
__global__ void k_shard_pick(int64_t n, const double* __restrict__ X, int64_t stride, const DavState* __restrict__ st,
                             double* __restrict__ send) {
  if (st->stop) return;
  const double* __restrict__ x = X + (int64_t)(st->m_next - 1) * stride;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) send[i] = x[i];
}
// local totals {|X_{m-1}|^2, X_v . A X_{m-1}} over this rank's rows into tot_out[0 .. MAXB] (zero beyond m): the dot-product
// pass of the single solver (split rows of the new sigma vector added on the way, as there)
template <int MV>
__global__ void __launch_bounds__(RED_T) k_shard_dots(int64_t n, const double* __restrict__ X, double* __restrict__ AX,
                                                      int64_t stride, double* __restrict__ partial, int width, unsigned* counter,
                                                      DavState* st, const SplitRows split, double* __restrict__ tot_out) {
  dots_eig_body<MV, false, true>(n, X, AX, stride, partial, width, counter, st, DavParams{}, split, blockIdx.x, gridDim.x, tot_out);
}
// the projected eigenproblem from the all-reduced totals: one wavefront
template <int MV>
__global__ void __launch_bounds__(64) k_shard_eig(DavState* st, const double* __restrict__ tot_in, const DavParams prm) {
  __shared__ double tot[MV + 1];
  __shared__ double sA[MV * MV], sM[MV * MV], sv_eig[MV + 1];
  __shared__ unsigned long long s_head[DAV_HEAD_WORDS];
  __shared__ double s_heff[MV * MV];
  if (st->stop) return;
  dav_state_prefetch<MV>(st, s_head, s_heff);
  if ((int)threadIdx.x < MV + 1) tot[threadIdx.x] = tot_in[threadIdx.x];
  wave_sync();
  wave_eig_step<MV>(st, s_head, s_heff, tot, prm, sA, sM, sv_eig);
}
// the residual / correction pass of a row-sharded solve: k_residual_precond's body, the local totals {|r|^2, |t|^2,
// X_v . t} folded by the workgroup that arrives last (fold_partials: the arithmetic of k_orth_dev's own fold)
template <int MV>
__global__ void __launch_bounds__(RED_T) k_shard_residual(int64_t n, double* __restrict__ X, const double* __restrict__ AX,
                                                          int64_t stride, const DavState* __restrict__ st,
                                                          const double* __restrict__ hdiag, const PenaltyDiag pd,
                                                          double* __restrict__ partial, int width, unsigned* counter,
                                                          double* __restrict__ tot_out) {
  residual_precond_body<MV, true>(n, X, AX, stride, st, hdiag, pd, partial, width, blockIdx.x, gridDim.x, counter, tot_out);
}
template <int MV>
__global__ void __launch_bounds__(RED_T) k_shard_orth(int64_t n, double* __restrict__ X, double* __restrict__ AX, int64_t stride,
                                                      DavState* st, const DavParams prm, const double* __restrict__ tot_in,
                                                      double* mail, long long seq, double* __restrict__ send) {
  orth_dev_body<MV>(n, X, AX, stride, st, prm, nullptr, 0, 0, mail, seq, blockIdx.x, gridDim.x, tot_in, send);
}

// ---- batched forms (sqd_solve_batch): blockIdx.z = subspace, one argument record per subspace in device memory.
// Every subspace runs on exactly the grid a single solve would give it (gb workgroups; the launch is sized for the
// largest), so its reductions fold the same partials in the same order: the same bits.
struct DavBatchArgs {
  int64_t n;           // D
  GPtr<double> X, AX;  // stride = n
  GPtr<double> partial, part_res;
  int width;
  unsigned gb;
  GPtr<unsigned> counter;
  GPtr<DavState> st;
  DavParams prm;
  SplitRows split;
  GPtr<const double> hdiag;
  PenaltyDiag pd;
  GPtr<double> mail;  // progress records (host-visible)
  GPtr<double> sol;
  GPtr<double> res;  // the run's outcome (host-visible)
  int dots_fused;    // k_sigma_dots_b has left this subspace's partials: k_dots_b returns at once
};
template <int MV>
__global__ void __launch_bounds__(RED_T, 4) k_dots_b(const DavBatchArgs* __restrict__ as) {
  const DavBatchArgs a = as[blockIdx.z];  // (a by-value copy: the record in SGPRs, as a kernel argument would be)
  if (blockIdx.x >= a.gb || a.dots_fused) return;
  dots_eig_body<MV, false>(a.n, a.X, a.AX, a.n, a.partial, a.width, a.counter, a.st, a.prm, a.split, blockIdx.x, a.gb);
}
// the element-gather class of a batch: sigma build + dot products of every such subspace in ONE launch (k_sigma_dots_eig
// without the eigen step, which k_eig_b does for all subspaces): blockIdx.z = member of the class, sub_of[z] = its record
template <int MV, bool SPIN>
__global__ void __launch_bounds__(RED_T, 4) k_sigma_dots_b(const DavBatchArgs* __restrict__ as, const DirectArgs* __restrict__ ds,
                                                            const int* __restrict__ sub_of) {
  __shared__ double red[16 * (MV + 1)];
  const DavBatchArgs a = as[sub_of[blockIdx.z]];
  const unsigned bx = blockIdx.x, nbx = a.gb;
  if (bx >= nbx) return;
  const DavState* st = a.st;
  if (st->stop) return;
  const DirectArgs dg = ds[blockIdx.z];
  const int nvec = st->m_next;
  const double* __restrict__ X = a.X;
  double* __restrict__ y = a.AX + (int64_t)(nvec - 1) * a.n;
  double acc[MV + 1];
#pragma unroll
  for (int v = 0; v < MV + 1; ++v) acc[v] = 0.0;
  if (nvec <= 2) dots_loop_direct<MV, (2 < MV ? 2 : MV), SPIN>(a.n, X, y, a.n, nvec, dg, bx, nbx, acc);
  else if (nvec <= 4) dots_loop_direct<MV, (4 < MV ? 4 : MV), SPIN>(a.n, X, y, a.n, nvec, dg, bx, nbx, acc);
  else if (nvec <= 8) dots_loop_direct<MV, (8 < MV ? 8 : MV), SPIN>(a.n, X, y, a.n, nvec, dg, bx, nbx, acc);
  else dots_loop_direct<MV, MV, SPIN>(a.n, X, y, a.n, nvec, dg, bx, nbx, acc);
  block_sum_multi<MV + 1>(acc, nvec + 1, red);
  if ((int)threadIdx.x < nvec + 1) a.partial[(int64_t)bx * a.width + threadIdx.x] = block_sum_multi_get<MV + 1>(red, threadIdx.x);
}
// the eigen step of every subspace of the batch: one workgroup each (fold of the partials k_dots_b left, then one wavefront)
template <int MV>
__global__ void __launch_bounds__(RED_T) k_eig_b(const DavBatchArgs* __restrict__ as) {
  __shared__ double tot[MV + 1];
  __shared__ double sA[MV * MV], sM[MV * MV], sv_eig[MV + 1];
  __shared__ unsigned long long s_head[DAV_HEAD_WORDS];
  __shared__ double s_heff[MV * MV];
  const DavBatchArgs a = as[blockIdx.x];
  DavState* st = a.st;
  if (st->stop) return;
  const int nvec = st->m_next;
  dav_state_prefetch<MV>(st, s_head, s_heff);
  fold_partials<false>(a.partial, (int)a.gb, a.width, nvec + 1, tot);
  if (threadIdx.x >= 64) return;
  wave_eig_step<MV>(st, s_head, s_heff, tot, a.prm, sA, sM, sv_eig);
}
// (four workgroups per CU: the 128-VGPR cap costs the <13> instantiation 4 spilled registers, 20 bytes of scratch, and is
// still 2-5 % faster per batched solve than three workgroups without a spill -- profiles/r06/resid_b_waves_probe.txt)
#ifndef SQD_RESID_B_WAVES
#define SQD_RESID_B_WAVES 4
#endif
template <int MV>
__global__ void __launch_bounds__(RED_T, SQD_RESID_B_WAVES) k_residual_precond_b(const DavBatchArgs* __restrict__ as) {
  const DavBatchArgs a = as[blockIdx.z];  // (a by-value copy: the record in SGPRs, as a kernel argument would be)
  if (blockIdx.x >= a.gb) return;
  residual_precond_body<MV>(a.n, a.X, a.AX, a.n, a.st, a.hdiag, a.pd, a.part_res, a.width, blockIdx.x, a.gb);
}
template <int MV>
__global__ void __launch_bounds__(RED_T) k_orth_dev_b(const DavBatchArgs* __restrict__ as, long long seq) {
  const DavBatchArgs a = as[blockIdx.z];  // (a by-value copy: the record in SGPRs, as a kernel argument would be)
  if (blockIdx.x >= a.gb) return;
  orth_dev_body<MV>(a.n, a.X, a.AX, a.n, a.st, a.prm, a.part_res, (int)a.gb, a.width, a.mail, seq, blockIdx.x, a.gb);
}
__global__ void k_solution_b(const DavBatchArgs* __restrict__ as) {
  const DavBatchArgs a = as[blockIdx.z];  // (a by-value copy: the record in SGPRs, as a kernel argument would be)
  if (blockIdx.x >= a.gb) return;
  solution_body(a.n, a.X, a.n, a.st, a.sol, a.res, blockIdx.x, a.gb);
}

// host side of a mailbox post: wait for sequence word `seq` in mailbox slot `slot`
static int wait_mail(sqd_ctx* c, int slot, long long seq) {
  return spin_wait_word(c->h_mail + (size_t)slot * MAIL_SLOT, seq, c->stream);
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
  SQD_TRY(wait_mail(c, 0, seq));
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

static int reserve_reduction_buffers(sqd_ctx* c) {
  SQD_TRY(c->partial.reserve((size_t)2 * RED_BLOCKS * (SQD_MAX_SPACE + 4) * 8));
  // scal: [0..128) scalars of stand-alone reductions | arrival counters | DavState
  const void* before = c->scal.p;
  // scal: ... | DavState | a second set of arrival counters (kernels with two "last workgroup" stages: k_observables)
  const size_t bytes = (size_t)128 * 8 + 3 * COUNT_WORDS * sizeof(unsigned) + ((sizeof(DavState) + 255) & ~size_t(255)) + 256;
  SQD_TRY(c->scal.reserve(bytes));
  if (c->scal.p != before)  // fresh allocation: the self-resetting arrival counters start from zero
    SQD_HIP_CHECK(hipMemsetAsync(c->scal.p, 0, bytes, c->stream));
  return SQD_OK;
}
unsigned* counter_ptr(sqd_ctx* c) { return reinterpret_cast<unsigned*>(c->scal.as<double>() + 128); }
int reserve_counters(sqd_ctx* c) { return reserve_reduction_buffers(c); }
static DavState* state_ptr_dev(sqd_ctx* c) {
  return reinterpret_cast<DavState*>(reinterpret_cast<char*>(counter_ptr(c)) + COUNT_WORDS * sizeof(unsigned));
}
void* dav_state_ptr(sqd_ctx* c) { return state_ptr_dev(c); }
unsigned* counter2_ptr(sqd_ctx* c) {
  return reinterpret_cast<unsigned*>(reinterpret_cast<char*>(state_ptr_dev(c)) + ((sizeof(DavState) + 255) & ~size_t(255)));
}
unsigned* counter3_ptr(sqd_ctx* c) { return counter2_ptr(c) + COUNT_WORDS; }

int dev_dot(sqd_ctx* c, const double* x, const double* y, double* out) {
  SQD_TRY(reserve_reduction_buffers(c));
  return multi_dot(c, x, 0, 1, y, out);
}

// pyscf get_init_guess (direct_spin1._get_init_guess): unit vector at the lowest diagonal element -- searched over
// the lower triangle A >= B when nelec_a == nelec_b and na == nb -- plus the +-1e-5 noise, normalised
static int enqueue_init_guess_impl(sqd_ctx* c, double* x, DavState* st, unsigned* counter) {
  const int64_t D = c->D;
  const unsigned gb = red_blocks(D);
  SQD_TRY(reserve_reduction_buffers(c));
  // the per-row minima of the diagonal (lower triangle only when pyscf's rule says so) were left by set_subspace
  // (k_tables_diag); every workgroup of k_init_guess finishes the argmin over them itself
  const double* pmin = c->guess_min.as<double>();
  const int64_t* pidx = reinterpret_cast<const int64_t*>(pmin + c->na);  // (unsharded: row range = all rows)
  hipLaunchKernelGGL(k_init_guess, dim3(gb), dim3(RED_T), 0, c->stream, D, pmin, pidx, (int)c->na, x, st, counter);
  SQD_HIP_CHECK(hipGetLastError());
  return SQD_OK;
}
int enqueue_init_guess(sqd_ctx* c, double* x) { return enqueue_init_guess_impl(c, x, nullptr, nullptr); }

// Residual threshold of a run.  Without a spin penalty: pyscf's rule, |r| < sqrt(tol) (lib.davidson1 as FCISolver.eig
// calls it; reference qiskit_addon_sqd/fermion.py:713-723) -- <c|H|c> is second order in the residual, 1e-9 Ha at the
// default tol, and a tighter rule buys accuracy the reference does not deliver (HF-centred 317^2: 29 instead of 40
// sigma builds, energies 3e-10 Ha apart).  With a penalty (fix_spin_) the returned <c|H|c> = Ritz value - shift *
// <penalty> is FIRST order in the residual: sqrt(tol)/32 keeps it inside 1e-6 Ha.  tol_residual overrides either.
static double residual_threshold(const sqd_davidson_opts* o) {
  if (o->tol_residual > 0.0) return o->tol_residual;
  return o->use_spin ? std::sqrt(o->tol) / 32.0 : std::sqrt(o->tol);
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
  const double toloose = residual_threshold(o);
  hipStream_t s = c->stream;
  const int nvecs = max_space + 1;
  SQD_TRY(c->X.reserve((size_t)nvecs * D * 8));
  SQD_TRY(c->AX.reserve((size_t)nvecs * D * 8));
  // an earlier asynchronous solve's state may still be leaving `sol` on the copy stream (k_state_copy): every writer
  // of `sol` waits for that copy first (a no-op behind sqd_solve's own buffer swap, which has checked the ticket)
  SQD_TRY(sol_writer_guard(c));
  SQD_TRY(c->sol.reserve((size_t)D * 8));
  SQD_TRY(reserve_reduction_buffers(c));
  double* X = c->X.as<double>();
  double* AX = c->AX.as<double>();
  const unsigned gb = red_blocks(D);
  const int width = SQD_MAX_SPACE + 4;
  unsigned* counter = counter_ptr(c);
  DavState* dst = state_ptr_dev(c);

  const bool timing = c->want_timing;
  if (timing) SQD_HIP_CHECK(hipEventRecord(c->ev[2], s));
  // ---- initial vector + state block
  // (a user vector is not normalised on the device: the first fused reduction measures |X_0|^2 and the
  // factor is carried in sv like that of every later basis vector)
  const bool guess_ready = (c->guess_x != nullptr && c->guess_x == X && !c->sharded());
  c->guess_x = nullptr;  // one-shot: the run overwrites X[0] and the state block
  c->dav_nvecs_hint = nvecs;
  if (ci0_host) {
    SQD_HIP_CHECK(hipMemcpyAsync(X, ci0_host, D * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_dav_init, dim3(1), dim3(256), 0, s, dst, counter);
    SQD_HIP_CHECK(hipGetLastError());
  } else if (!guess_ready) {  // (else: set_subspace's last launch already did both)
    SQD_TRY(enqueue_init_guess_impl(c, X, dst, counter));
  }

  DavParams prm;
  prm.tol = o->tol;
  prm.tol2 = toloose * toloose;
  prm.lindep = o->lindep;
  prm.max_space = max_space;
  PenaltyDiag pd;
  {
    int form = o->use_spin;
    // pyscf SpinPenaltyFCISolver.contract_2e chooses the form with sz = |neleca - nelecb| / 2
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
  double* mail_prog = c->d_mail + MAIL_SLOT;
  const double* h_prog = c->h_mail + MAIL_SLOT;
  double* part_res = c->partial.as<double>() + (size_t)RED_BLOCKS * width;  // residual / orth: a buffer of their own

  // every time_sigma_every-th sigma launch of this context is bracketed by events (stats->ms_sigma); an
  // event pair costs ~10 us of stream time, so this is sampling, and off unless asked for
  const int ev_every = o->time_sigma_every > 0 ? o->time_sigma_every : 0;
  const int max_ev = ev_every ? (int)c->sig_ev.size() / 4 : 0;
  int nev = 0;
  c->dav_ev_iter.clear();

  // sigma reads "which vector" from the state block: X[m_next - 1] -> AX[m_next - 1]
  struct IndexGuard {
    sqd_ctx* c;
    ~IndexGuard() {
      c->sigma_stop = nullptr;
      c->sigma_index = nullptr;
      c->sigma_defer_reduce = false;
    }
  } guard{c};
  c->sigma_stop = &dst->stop;
  c->sigma_index = &dst->m_next;
  // split rows are summed by k_dots_eig (the squared-penalty form chains three sigma launches through scratch
  // vectors and keeps its reduce launches)
  int form_sel = o->use_spin;
  if (form_sel == 3) {
    const double szh = 0.5 * std::abs(c->nelec[0] - c->nelec[1]);
    form_sel = (o->ss < szh * (szh + 1.0) + 0.1) ? 1 : 2;
  }
  SplitRows split{nullptr, nullptr, c->nb};
  // (the whole-row opposite-spin kernel of sqd_opp.hip -- plain operator on the subspaces of the sparse-product path --
  // writes complete rows: nothing to add)
  if (c->sig_opp && form_sel != 2) {  // (plain operator and the linear penalty form: both by whole rows)
    const int32_t* ri = nullptr;
    const double* pp = nullptr;
    if (opp_split(c, &ri, &pp)) {
      c->sigma_defer_reduce = true;
      split.rowinfo = ri;
      split.partial = pp;
    }
  } else if (c->n_multi > 0 && form_sel != 2 && !c->sig_direct) {
    c->sigma_defer_reduce = true;
    split.rowinfo = c->rowinfo.as<int32_t>();
    split.partial = c->sig_partial.as<double>();
  }

  // Host loop.  A round = the sigma build (part A) + the three BLAS-1 launches behind it (part B); the device decides
  // everything, the host only keeps the queue fed and watches the progress records.  How far ahead it enqueues:
  //  * from round FULL_FROM on, one WHOLE round is queued behind the running one: no launch ever waits for the host,
  //    and a stop wastes one round of early-exit launches (4 dependent dispatches, ~10 us) -- nothing on a long solve;
  //  * in the first rounds only the NEXT SIGMA BUILD is queued ahead: part B of round r + 1 is enqueued when the
  //    record of round r has been seen, while that sigma build runs (5 us at the headline, 27 us HF-centred: time for
  //    the three launches).  A stop then wastes one dispatch instead of four -- the solves of uniform-random sets
  //    converge in 2-3 rounds, and 8 us are 5 % of them.
  const bool lockstep = o->verbose != 0;  // verbose: round by round, each record printed before the next round starts
  const int full_from = 3;
  long long seq_of[4] = {0, 0, 0, 0};  // sequence numbers of the latest rounds enqueued (ring)
  bool stopped = false;
  auto settle = [&](int j) -> int {  // wait for round j's progress record
    SQD_TRY(wait_mail(c, 1, seq_of[j & 3]));
    if (h_prog[MAIL_PAYLOAD + 1] != 0.0) stopped = true;
    if (o->verbose)
      std::fprintf(stderr, "[sqd davidson] it %d space %d e %.12f de %.3e |r| %.3e\n", (int)h_prog[MAIL_PAYLOAD + 0] - 1,
                   (int)h_prog[MAIL_PAYLOAD + 5], h_prog[MAIL_PAYLOAD + 2], h_prog[MAIL_PAYLOAD + 3],
                   std::sqrt(h_prog[MAIL_PAYLOAD + 4]));
    return SQD_OK;
  };
  static const int64_t dots_split_d = [] {  // tuning hook: subspace dimension from which dots and eigen step are two launches
    const char* env = std::getenv("SQD_DOTS_SPLIT_D");
    return env ? (int64_t)std::atoll(env) : (int64_t)200000;  // (profiles/r05/dots_split_probe.txt)
  }();
  const bool split_dots = D >= dots_split_d;
  // element-gather sigma, fused reduction (not from D = dots_split_d on, not the squared penalty form): the sigma build and
  // the reduction behind it are ONE launch, k_sigma_dots_eig -- except a sigma launch that is being timed, which goes out as
  // the sigma kernel alone between its events and the reduction behind them (the same bits either way)
  bool fuse_direct = c->sig_direct && c->sig_rows == 0 && !split_dots && form_sel != 2 && max_space <= 12 && !c->sharded();
  if (fuse_direct)
    if (const char* env = std::getenv("SQD_DAV_FUSE_DIRECT")) fuse_direct = std::atoi(env) != 0;  // test / probe hook
  DirectArgs dg;
  if (fuse_direct) fill_direct_args(c, X, AX, 0, form_sel == 1, o->ss, o->shift, D, D, &dg);
  bool fused_of[4] = {false, false, false, false};  // was round r's sigma build the fused launch? (ring, like seq_of)
  auto part_a = [&](int round) -> int {  // the sigma build of a round (reads "which vector" from the state block)
    fused_of[round & 3] = false;
    if (fuse_direct && !(ev_every && nev < max_ev && (c->sigma_launches % ev_every == 0))) {
      fused_of[round & 3] = true;
      ++c->sigma_launches;
      if (form_sel == 1)
        hipLaunchKernelGGL((k_sigma_dots_eig<13, true>), dim3(gb), dim3(RED_T), 0, s, D, (const double*)X, AX, D,
                           c->partial.as<double>(), width, counter, dst, prm, dg);
      else
        hipLaunchKernelGGL((k_sigma_dots_eig<13, false>), dim3(gb), dim3(RED_T), 0, s, D, (const double*)X, AX, D,
                           c->partial.as<double>(), width, counter, dst, prm, dg);
      SQD_HIP_CHECK(hipGetLastError());
      return SQD_OK;
    }
    const bool timed = ev_every && nev < max_ev && (c->sigma_launches % ev_every == 0);
    ++c->sigma_launches;
    if (timed) {
      SQD_HIP_CHECK(hipEventRecord(c->sig_ev[4 * nev], s));
      c->ev_after_sigma_kernel = c->sig_ev[4 * nev + 1];  // recorded by launch_sigma right after k_sigma
    }
    const int rc_h = apply_h(c, X, AX, o->use_spin, o->ss, o->shift, D, D);
    c->ev_after_sigma_kernel = nullptr;
    SQD_TRY(rc_h);
    if (timed) {
      SQD_HIP_CHECK(hipEventRecord(c->sig_ev[4 * nev + 2], s));
      // an EMPTY bracket right behind: what two event records cost by themselves at this point of the stream
      SQD_HIP_CHECK(hipEventRecord(c->sig_ev[4 * nev + 3], s));
      c->dav_ev_iter.push_back(round);
      ++nev;
    }
    return SQD_OK;
  };
  auto part_b = [&](int round) -> int {
    const long long seq = ++c->mail_seq;
    seq_of[round & 3] = seq;
    if (max_space <= 12) {
      if (split_dots) {
        hipLaunchKernelGGL((k_dots_s<13>), dim3(gb), dim3(RED_T), 0, s, D, (const double*)X, AX, D,
                           c->partial.as<double>(), width, counter, dst, prm, split);
        hipLaunchKernelGGL((k_eig_s<13>), dim3(1), dim3(RED_T), 0, s, (const double*)c->partial.as<double>(), (int)gb, width, dst,
                           prm);
      } else if (!fused_of[round & 3]) {  // (fused: part A has done it)
        hipLaunchKernelGGL((k_dots_eig<13>), dim3(gb), dim3(RED_T), 0, s, D, (const double*)X, AX, D,
                           c->partial.as<double>(), width, counter, dst, prm, split);
      }
      hipLaunchKernelGGL((k_residual_precond<13>), dim3(gb), dim3(RED_T), 0, s, D, X, (const double*)AX, D,
                         (const DavState*)dst, (const double*)c->hdiag.as<double>(), pd, part_res, width);
      hipLaunchKernelGGL((k_orth_dev<13>), dim3(gb), dim3(RED_T), 0, s, D, X, AX, D, dst, prm, (const double*)part_res,
                         (int)gb, width, mail_prog, seq, c->sol.as<double>(), c->d_mail + 2 * MAIL_SLOT + MAIL_PAYLOAD);
    } else {
      hipLaunchKernelGGL((k_dots_eig<MAXB>), dim3(gb), dim3(RED_T), 0, s, D, (const double*)X, AX, D,
                         c->partial.as<double>(), width, counter, dst, prm, split);
      hipLaunchKernelGGL((k_residual_precond<MAXB>), dim3(gb), dim3(RED_T), 0, s, D, X, (const double*)AX, D,
                         (const DavState*)dst, (const double*)c->hdiag.as<double>(), pd, part_res, width);
      hipLaunchKernelGGL((k_orth_dev<MAXB>), dim3(gb), dim3(RED_T), 0, s, D, X, AX, D, dst, prm, (const double*)part_res,
                         (int)gb, width, mail_prog, seq, c->sol.as<double>(), c->d_mail + 2 * MAIL_SLOT + MAIL_PAYLOAD);
    }
    SQD_HIP_CHECK(hipGetLastError());
    // (The launch that stops the solve forms the solution itself -- k_orth_dev, sol_out: rounds 2-5 queued a conditional
    // k_solution behind each of the first rounds instead, and every round that did not stop paid 4.7 us for its early exit.)
    return SQD_OK;
  };
  int last_settled = -1;
  {
    const int max_rounds = o->max_cycle;
    int n_a = 0, n_b = 0;  // rounds whose part A / part B have been enqueued
    SQD_TRY(part_a(n_a++));
    SQD_TRY(part_b(n_b++));
    for (int r = 0;;) {  // r = the round whose record is waited for next
      if (!lockstep) {
        if (n_a == r + 1 && n_a < max_rounds) SQD_TRY(part_a(n_a++));
        if (n_a == r + 2 && n_b == r + 1 && r + 1 >= full_from) SQD_TRY(part_b(n_b++));
      }
      SQD_TRY(settle(r));
      last_settled = r;
      if (stopped) break;
      ++r;
      // round r has to be complete in the queue before its record can be waited for
      if (n_a == r && n_a < max_rounds) SQD_TRY(part_a(n_a++));
      if (n_b == r && n_b < n_a) SQD_TRY(part_b(n_b++));
      if (n_b == r) break;  // the cycle limit: nothing more to run
    }
  }
  // solution = Ritz vector of the last projected problem, normalised; the run's outcome to mailbox slot 2 -- unless the
  // conditional launch behind the stopping round has formed it already
  // (WHICH round stopped the solve is read from the record itself -- its iteration counter freezes at the stop --: with
  // rounds queued ahead, the record the host finds may already be a later round's)
  (void)last_settled;
  // ... unless the launch that stopped the run has formed it (record word 6; a later round's record, found when rounds
  // were queued ahead, repeats it from the state block).  Not formed: the cycle limit, a stop taken inside the eigen step
  // (linear dependence, a zero start vector)
  if (!(stopped && h_prog[MAIL_PAYLOAD + 6] != 0.0)) {
    hipLaunchKernelGGL(k_solution, dim3(gb), dim3(RED_T), 0, s, D, (const double*)X, D, (const DavState*)dst,
                       c->sol.as<double>(), c->d_mail + 2 * MAIL_SLOT + MAIL_PAYLOAD, 0);
    SQD_HIP_CHECK(hipGetLastError());
  }
  if (timing) SQD_HIP_CHECK(hipEventRecord(c->ev[3], s));
  c->dav_timed = timing;
  c->have_solution = true;
  c->dav_nev = nev;
  if (st) std::memset(st, 0, sizeof(*st));
  if (defer_sync) return SQD_OK;
  SQD_STREAM_SYNC(s);
  return davidson_collect(c, st);
}

// ---- batched Davidson (sqd_solve_batch): the same four launches per round advance EVERY subspace of the batch.  Each
// subspace has its own state block, arrival counters, partial arrays and mailbox (its sub-context's), stops itself, and
// from then on its workgroups return at once; the host keeps rounds coming until every subspace has stopped.
size_t davidson_batch_bytes(size_t nsub) { return nsub * (sizeof(DavBatchArgs) + 64 + sizeof(int)) + sigma_batch_bytes(nsub) + 512; }

static long long common_seq(const std::vector<sqd_ctx*>& subs) {
  // one sequence number for the same mailbox word of every subspace: above everything any of them has used
  int64_t s = 0;
  for (sqd_ctx* c : subs) s = c->mail_seq > s ? c->mail_seq : s;
  ++s;
  for (sqd_ctx* c : subs) c->mail_seq = s;
  return (long long)s;
}

int davidson_batch_prepare(sqd_ctx* parent, const std::vector<sqd_ctx*>& subs, const sqd_davidson_opts* o, char* h,
                           char* d, size_t* off_io, DavBatchPlan* plan) {
  const int n = (int)subs.size();
  int max_space = o->max_space;
  if (max_space < 2) max_space = 2;
  if (max_space > SQD_MAX_SPACE) max_space = SQD_MAX_SPACE;
  const double toloose = residual_threshold(o);
  const int nvecs = max_space + 1;
  const int width = SQD_MAX_SPACE + 4;
  plan->max_space = max_space;
  plan->max_cycle = o->max_cycle;
  plan->n = n;
  plan->gb = 1;
  size_t off = (*off_io + 63) & ~size_t(63);
  DavBatchArgs* ha = reinterpret_cast<DavBatchArgs*>(h + off);
  plan->args = d + off;
  off += (size_t)n * sizeof(DavBatchArgs);
  std::vector<const double*> xin(n);
  std::vector<double*> axout(n);
  int form = o->use_spin;
  for (int p = 0; p < n; ++p) {
    sqd_ctx* c = subs[p];
    const int64_t D = c->D;
    SQD_TRY(c->X.reserve((size_t)nvecs * D * 8));
    SQD_TRY(c->AX.reserve((size_t)nvecs * D * 8));
    SQD_TRY(c->sol.reserve((size_t)D * 8));
    SQD_TRY(reserve_reduction_buffers(c));
    double* X = c->X.as<double>();
    DavState* dst = state_ptr_dev(c);
    unsigned* counter = counter_ptr(c);
    const bool guess_ready = (c->guess_x != nullptr && c->guess_x == X);
    c->guess_x = nullptr;
    c->dav_nvecs_hint = nvecs;
    if (!guess_ready) SQD_TRY(enqueue_init_guess_impl(c, X, dst, counter));  // (workspace moved: rare)
    DavBatchArgs& a = ha[p];
    a.n = D;
    a.X = X;
    a.AX = c->AX.as<double>();
    a.partial = c->partial.as<double>();
    a.part_res = c->partial.as<double>() + (size_t)RED_BLOCKS * width;
    a.width = width;
    a.gb = red_blocks(D);
    plan->gb = a.gb > plan->gb ? a.gb : plan->gb;
    a.counter = counter;
    a.st = dst;
    a.prm.tol = o->tol;
    a.prm.tol2 = toloose * toloose;
    a.prm.lindep = o->lindep;
    a.prm.max_space = max_space;
    {
      int f = o->use_spin;
      const double szh = 0.5 * std::abs(c->nelec[0] - c->nelec[1]);
      if (f == 3) f = (o->ss < szh * (szh + 1.0) + 0.1) ? 1 : 2;
      if (f == 2) {
        set_error("the squared spin penalty is not available in a batched solve");
        return SQD_ERR_STATE;
      }
      form = f;
      const double sz = 0.5 * (c->nelec[0] - c->nelec[1]);
      a.pd.form = f;
      a.pd.shift = o->shift;
      a.pd.ss = o->ss;
      a.pd.szterm = sz * (sz + 1.0);
      a.pd.sa = c->sp[0].strs.as<uint64_t>();
      a.pd.sb = c->sp[1].strs.as<uint64_t>();
      a.pd.nb = c->nb;
    }
    a.hdiag = c->hdiag.as<double>();
    a.mail = c->d_mail + MAIL_SLOT;
    a.sol = c->sol.as<double>();
    a.res = c->d_mail + 2 * MAIL_SLOT + MAIL_PAYLOAD;
    a.split = SplitRows{nullptr, nullptr, c->nb};
    // sigma reads "which vector" and "whether to stop" from the state block; split rows are added by k_dots_eig
    c->sigma_stop = &dst->stop;
    c->sigma_index = &dst->m_next;
    c->sigma_defer_reduce = false;
    if (c->n_multi > 0 && !c->sig_direct) {
      c->sigma_defer_reduce = true;
      a.split.rowinfo = c->rowinfo.as<int32_t>();
      a.split.partial = c->sig_partial.as<double>();
    }
    xin[p] = X;
    axout[p] = c->AX.as<double>();
    c->dav_timed = false;
    c->dav_nev = 0;
    c->dav_ev_iter.clear();
    c->have_solution = true;
  }
  // the element-gather class: sigma build and dot products in one launch, as the single solve has them (k_sigma_dots_eig)
  plan->n_fused = 0;
  bool fuse = max_space <= 12;
  if (fuse)
    if (const char* env = std::getenv("SQD_DAV_FUSE_DIRECT")) fuse = std::atoi(env) != 0;  // test / probe hook
  for (int p = 0; p < n; ++p) ha[p].dots_fused = 0;
  if (fuse) {
    std::vector<int> fidx;
    for (int p = 0; p < n; ++p)
      if (subs[p]->sig_direct && subs[p]->sig_rows == 0) fidx.push_back(p);
    if (!fidx.empty()) {
      off = (off + 63) & ~size_t(63);
      DirectArgs* hd = reinterpret_cast<DirectArgs*>(h + off);
      plan->fused_args = d + off;
      off += fidx.size() * sizeof(DirectArgs);
      off = (off + 63) & ~size_t(63);
      int* hm = reinterpret_cast<int*>(h + off);
      plan->fused_map = d + off;
      off += fidx.size() * sizeof(int);
      for (size_t k = 0; k < fidx.size(); ++k) {
        sqd_ctx* c = subs[fidx[k]];
        fill_direct_args(c, xin[fidx[k]], axout[fidx[k]], 0, form == 1, o->ss, o->shift, c->D, c->D, &hd[k]);
        hm[k] = fidx[k];
        ha[fidx[k]].dots_fused = 1;
      }
      plan->n_fused = (int)fidx.size();
      plan->fused_spin = (form == 1);
    }
  }
  const int rc = sigma_batch_plan(subs, xin, axout, 0, form == 1, o->ss, o->shift, 1, h, d, &off, &plan->sigma,
                                  /*skip_direct=*/plan->n_fused > 0);
  for (sqd_ctx* c : subs) {
    c->sigma_stop = nullptr;
    c->sigma_index = nullptr;
    c->sigma_defer_reduce = false;
  }
  SQD_TRY(rc);
  *off_io = off;
  return SQD_OK;
}

int davidson_batch_run(sqd_ctx* parent, const std::vector<sqd_ctx*>& subs, const DavBatchPlan& plan) {
  hipStream_t s = parent->stream;
  const int n = plan.n;
  const DavBatchArgs* args = reinterpret_cast<const DavBatchArgs*>(plan.args);
  const dim3 grid(plan.gb, 1, (unsigned)n);
  long long seq_of[4] = {0, 0, 0, 0};
  bool stopped = false;
  auto settle = [&](int j) -> int {  // round j's progress record of every subspace
    bool all = true;
    for (sqd_ctx* c : subs) {
      SQD_TRY(spin_wait_word(c->h_mail + MAIL_SLOT, seq_of[j & 3], s));
      if (c->h_mail[MAIL_SLOT + MAIL_PAYLOAD + 1] == 0.0) all = false;
    }
    stopped = all;
    return SQD_OK;
  };
  auto part_a = [&]() -> int {
    if (plan.n_fused > 0) {
      const dim3 fg(plan.gb, 1, (unsigned)plan.n_fused);
      const DirectArgs* ds = reinterpret_cast<const DirectArgs*>(plan.fused_args);
      const int* sub_of = reinterpret_cast<const int*>(plan.fused_map);
      if (plan.fused_spin) hipLaunchKernelGGL((k_sigma_dots_b<13, true>), fg, dim3(RED_T), 0, s, args, ds, sub_of);
      else hipLaunchKernelGGL((k_sigma_dots_b<13, false>), fg, dim3(RED_T), 0, s, args, ds, sub_of);
      SQD_HIP_CHECK(hipGetLastError());
    }
    return sigma_batch_launch(parent, plan.sigma);
  };
  auto part_b = [&](int round) -> int {
    const long long seq = common_seq(subs);
    seq_of[round & 3] = seq;
    if (plan.max_space <= 12) {
      hipLaunchKernelGGL((k_dots_b<13>), grid, dim3(RED_T), 0, s, args);
      hipLaunchKernelGGL((k_eig_b<13>), dim3(grid.z), dim3(RED_T), 0, s, args);
      hipLaunchKernelGGL((k_residual_precond_b<13>), grid, dim3(RED_T), 0, s, args);
      hipLaunchKernelGGL((k_orth_dev_b<13>), grid, dim3(RED_T), 0, s, args, seq);
    } else {
      hipLaunchKernelGGL((k_dots_b<MAXB>), grid, dim3(RED_T), 0, s, args);
      hipLaunchKernelGGL((k_eig_b<MAXB>), dim3(grid.z), dim3(RED_T), 0, s, args);
      hipLaunchKernelGGL((k_residual_precond_b<MAXB>), grid, dim3(RED_T), 0, s, args);
      hipLaunchKernelGGL((k_orth_dev_b<MAXB>), grid, dim3(RED_T), 0, s, args, seq);
    }
    SQD_HIP_CHECK(hipGetLastError());
    return SQD_OK;
  };
  {
    // the single solve's enqueue-ahead policy (run_davidson): the next sigma build always, a whole round from the
    // fourth round on
    const int max_rounds = plan.max_cycle, full_from = 3;
    int n_a = 0, n_b = 0;
    SQD_TRY(part_a());
    ++n_a;
    SQD_TRY(part_b(n_b++));
    for (int r = 0;;) {
      if (n_a == r + 1 && n_a < max_rounds) {
        SQD_TRY(part_a());
        ++n_a;
      }
      if (n_a == r + 2 && n_b == r + 1 && r + 1 >= full_from) SQD_TRY(part_b(n_b++));
      SQD_TRY(settle(r));
      if (stopped) break;
      ++r;
      if (n_a == r && n_a < max_rounds) {
        SQD_TRY(part_a());
        ++n_a;
      }
      if (n_b == r && n_b < n_a) SQD_TRY(part_b(n_b++));
      if (n_b == r) break;  // the cycle limit
    }
  }
  hipLaunchKernelGGL(k_solution_b, grid, dim3(RED_T), 0, s, args);
  SQD_HIP_CHECK(hipGetLastError());
  return SQD_OK;
}

// ---- host side of the row-sharded Davidson: one call per stage of an iteration; the caller (sharded.py) puts its
// collectives between them on the same stream.  No host wait except in shard_dav_orth (the stop decision).
static int shard_check(sqd_ctx* c) {
  if (!c->have_subspace) {
    set_error("no subspace set");
    return SQD_ERR_STATE;
  }
  if (!c->shard_active) {
    set_error("no row-sharded Davidson run in progress (sqd_shard_dav_begin)");
    return SQD_ERR_STATE;
  }
  return SQD_OK;
}
int shard_dav_begin(sqd_ctx* c, const sqd_davidson_opts* o, double** d_x0) {
  if (!c->have_subspace) {
    set_error("no subspace set");
    return SQD_ERR_STATE;
  }
  int max_space = o->max_space;
  if (max_space < 2) max_space = 2;
  if (max_space > SQD_MAX_SPACE) max_space = SQD_MAX_SPACE;
  int form = o->use_spin;
  const double szh = 0.5 * std::abs(c->nelec[0] - c->nelec[1]);
  if (form == 3) form = (o->ss < szh * (szh + 1.0) + 0.1) ? 1 : 2;
  if (form == 2) {
    set_error("the squared spin penalty needs a second all-gather per sigma build: not available in the native sharded run");
    return SQD_ERR_INVALID;
  }
  const int64_t Dl = (c->row1 - c->row0) * c->nb;
  const int nvecs = max_space + 1;
  SQD_TRY(c->X.reserve((size_t)nvecs * Dl * 8));
  SQD_TRY(c->AX.reserve((size_t)nvecs * Dl * 8));
  SQD_TRY(sol_writer_guard(c));
  SQD_TRY(c->sol.reserve((size_t)Dl * 8));
  SQD_TRY(c->tmp1.reserve((size_t)Dl * 8));  // the send buffer of the all-gather
  SQD_TRY(reserve_reduction_buffers(c));
  SQD_TRY(c->shard_tot.reserve((size_t)2 * (MAXB + 2) * 8));
  c->guess_x = nullptr;
  const double toloose = residual_threshold(o);
  c->shard_prm_tol = o->tol;
  c->shard_prm_tol2 = toloose * toloose;
  c->shard_prm_lindep = o->lindep;
  c->shard_max_space = max_space;
  c->shard_form = form;
  c->shard_ss = o->ss;
  c->shard_shift = o->shift;
  c->shard_Dl = Dl;
  c->shard_active = true;
  c->shard_send_fresh = false;
  c->have_solution = false;
  hipLaunchKernelGGL(k_dav_init, dim3(1), dim3(256), 0, c->stream, state_ptr_dev(c), counter_ptr(c));
  SQD_HIP_CHECK(hipGetLastError());
  *d_x0 = c->X.as<double>();
  return SQD_OK;
}
static DavParams shard_params(const sqd_ctx* c) {
  DavParams prm;
  prm.tol = c->shard_prm_tol;
  prm.tol2 = c->shard_prm_tol2;
  prm.lindep = c->shard_prm_lindep;
  prm.max_space = c->shard_max_space;
  return prm;
}
int shard_dav_pick(sqd_ctx* c, double** d_send) {
  SQD_TRY(shard_check(c));
  const int64_t Dl = c->shard_Dl;
  if (!c->shard_send_fresh) {  // the start vector; every later one is put there by the orth stage that forms it
    hipLaunchKernelGGL(k_shard_pick, dim3(red_blocks(Dl)), dim3(RED_T), 0, c->stream, Dl, (const double*)c->X.as<double>(), Dl,
                       (const DavState*)state_ptr_dev(c), c->tmp1.as<double>());
    SQD_HIP_CHECK(hipGetLastError());
  }
  c->shard_send_fresh = false;
  *d_send = c->tmp1.as<double>();
  return SQD_OK;
}
// split rows of the sigma vector are summed by the dots stage (as in the single solver: no k_sigma_reduce launch) unless the
// squared-penalty form chains several sigma launches through scratch vectors
// (a "shard" that holds all rows -- a group of one -- may run the whole-row opposite-spin kernel of sqd_opp.hip for the
// plain operator: its rows in several pieces are then the split rows, not the work items')
static bool shard_split_rows(const sqd_ctx* c, const int32_t** rowinfo, const double** partial) {
  if (c->sig_opp && c->shard_form != 2) return opp_split(c, rowinfo, partial);
  if (!(c->n_multi > 0 && c->shard_form != 2 && !c->sig_direct && c->sig_rows == 0)) return false;
  *rowinfo = c->rowinfo.as<int32_t>();
  *partial = c->sig_partial.as<double>();
  return true;
}
static bool shard_defers_reduce(const sqd_ctx* c) {
  const int32_t* ri = nullptr;
  const double* pp = nullptr;
  return shard_split_rows(c, &ri, &pp);
}
// part 0: the whole sigma build on the gathered vector.  Parts 1 and 2 (both called, in this order, around the
// all-gather): 1 needs only this rank's rows of the vector -- they lie in the send buffer the pick stage returned -- and
// runs WHILE the gather is in flight: the own-row work items (diagonal, beta links, beta singles x alpha occupation)
// without their folded alpha links; 2, behind the gather, everything that reads other ranks' rows.  The squared
// penalty form chains three operators through scratch vectors and takes part 2 whole.  Same bits as part 0.
int shard_dav_sigma(sqd_ctx* c, const double* d_full, int part) {
  SQD_TRY(shard_check(c));
  if (part < 0 || part > 2) {
    set_error("shard_dav_sigma: part must be 0, 1 or 2");
    return SQD_ERR_INVALID;
  }
  const bool splittable = c->shard_form != 2;
  if (part == 1 && !splittable) return SQD_OK;
  DavState* dst = state_ptr_dev(c);
  c->sigma_stop = &dst->stop;
  c->sigma_index = &dst->m_next;
  c->sigma_defer_reduce = shard_defers_reduce(c);
  c->sig_part = splittable ? part : 0;
  c->sig_c_own = (part == 1) ? c->tmp1.as<double>() : nullptr;
  const int rc = apply_h(c, d_full, c->AX.as<double>(), c->shard_form, c->shard_ss, c->shard_shift, 0, c->shard_Dl);
  c->sig_part = 0;
  c->sig_c_own = nullptr;
  c->sigma_stop = nullptr;
  c->sigma_index = nullptr;
  c->sigma_defer_reduce = false;
  return rc;
}
int shard_dav_dots(sqd_ctx* c, double** d_tot, int* count) {
  SQD_TRY(shard_check(c));
  const int64_t Dl = c->shard_Dl;
  const unsigned gb = red_blocks(Dl);
  const int width = SQD_MAX_SPACE + 4;
  double* tot = c->shard_tot.as<double>();
  SplitRows split{nullptr, nullptr, c->nb};
  {  // (the sigma stage left split rows in pieces: this pass adds them, as k_dots_eig does)
    const int32_t* ri = nullptr;
    const double* pp = nullptr;
    if (shard_split_rows(c, &ri, &pp)) {
      split.rowinfo = ri;
      split.partial = pp;
    }
  }
  if (c->shard_max_space <= 12)
    hipLaunchKernelGGL((k_shard_dots<13>), dim3(gb), dim3(RED_T), 0, c->stream, Dl, (const double*)c->X.as<double>(),
                       c->AX.as<double>(), Dl, c->partial.as<double>(), width, counter_ptr(c), state_ptr_dev(c), split, tot);
  else
    hipLaunchKernelGGL((k_shard_dots<MAXB>), dim3(gb), dim3(RED_T), 0, c->stream, Dl, (const double*)c->X.as<double>(),
                       c->AX.as<double>(), Dl, c->partial.as<double>(), width, counter_ptr(c), state_ptr_dev(c), split, tot);
  SQD_HIP_CHECK(hipGetLastError());
  *d_tot = tot;
  *count = MAXB + 1;
  return SQD_OK;
}
int shard_dav_residual(sqd_ctx* c, double** d_tot2, int* count, bool eig_done) {
  SQD_TRY(shard_check(c));
  const int64_t Dl = c->shard_Dl;
  const unsigned gb = red_blocks(Dl);
  const int width = SQD_MAX_SPACE + 4;
  DavState* dst = state_ptr_dev(c);
  const DavParams prm = shard_params(c);
  double* tot = c->shard_tot.as<double>();
  double* tot2 = tot + (MAXB + 2);
  double* part_res = c->partial.as<double>() + (size_t)RED_BLOCKS * width;
  PenaltyDiag pd;
  {
    const double sz = 0.5 * (c->nelec[0] - c->nelec[1]);
    pd.form = c->shard_form;
    pd.shift = c->shard_shift;
    pd.ss = c->shard_ss;
    pd.szterm = sz * (sz + 1.0);
    pd.sa = c->sp[0].strs.as<uint64_t>() + c->row0;  // (element i of the shard lies in row row0 + i / nb)
    pd.sb = c->sp[1].strs.as<uint64_t>();
    pd.nb = c->nb;
  }
  if (c->shard_max_space <= 12) {
    if (!eig_done) hipLaunchKernelGGL((k_shard_eig<13>), dim3(1), dim3(64), 0, c->stream, dst, (const double*)tot, prm);
    hipLaunchKernelGGL((k_shard_residual<13>), dim3(gb), dim3(RED_T), 0, c->stream, Dl, c->X.as<double>(),
                       (const double*)c->AX.as<double>(), Dl, (const DavState*)dst, (const double*)c->hdiag.as<double>(), pd,
                       part_res, width, counter_ptr(c), tot2);
  } else {
    if (!eig_done) hipLaunchKernelGGL((k_shard_eig<MAXB>), dim3(1), dim3(64), 0, c->stream, dst, (const double*)tot, prm);
    hipLaunchKernelGGL((k_shard_residual<MAXB>), dim3(gb), dim3(RED_T), 0, c->stream, Dl, c->X.as<double>(),
                       (const double*)c->AX.as<double>(), Dl, (const DavState*)dst, (const double*)c->hdiag.as<double>(), pd,
                       part_res, width, counter_ptr(c), tot2);
  }
  SQD_HIP_CHECK(hipGetLastError());
  *d_tot2 = tot2;
  *count = MAXB + 2;
  return SQD_OK;
}
int shard_dav_orth(sqd_ctx* c, long long* seq_out) {
  SQD_TRY(shard_check(c));
  const int64_t Dl = c->shard_Dl;
  const unsigned gb = red_blocks(Dl);
  DavState* dst = state_ptr_dev(c);
  const DavParams prm = shard_params(c);
  const double* tot2 = c->shard_tot.as<double>() + (MAXB + 2);
  const long long seq = ++c->mail_seq;
  double* mail_prog = c->d_mail + MAIL_SLOT;
  if (c->shard_max_space <= 12)
    hipLaunchKernelGGL((k_shard_orth<13>), dim3(gb), dim3(RED_T), 0, c->stream, Dl, c->X.as<double>(), c->AX.as<double>(), Dl,
                       dst, prm, tot2, mail_prog, seq, c->tmp1.as<double>());
  else
    hipLaunchKernelGGL((k_shard_orth<MAXB>), dim3(gb), dim3(RED_T), 0, c->stream, Dl, c->X.as<double>(), c->AX.as<double>(), Dl,
                       dst, prm, tot2, mail_prog, seq, c->tmp1.as<double>());
  SQD_HIP_CHECK(hipGetLastError());
  c->shard_send_fresh = true;  // (the orth stage left the next vector in the send buffer)
  if (seq_out) *seq_out = seq;
  return SQD_OK;
}
// A whole iteration in ONE call for a group of one rank (every collective is the identity: nothing has to happen between
// the stages): pick -> sigma on the send buffer (= the whole vector) -> the single solver's fused dots + eigen kernel (no
// totals to all-reduce, so the workgroup that arrives last goes on to the projected problem: one launch less) -> residual ->
// orth.  Same stages, same bits as the five calls.
int shard_dav_iteration(sqd_ctx* c, long long* seq_out) {
  SQD_TRY(shard_check(c));
  if (c->row0 != 0 || c->row1 != c->na) {
    set_error("shard_dav_iteration: the context holds a true row shard (the stages need their collectives)");
    return SQD_ERR_STATE;
  }
  double* send = nullptr;
  SQD_TRY(shard_dav_pick(c, &send));
  SQD_TRY(shard_dav_sigma(c, send, 0));
  {
    const int64_t Dl = c->shard_Dl;
    const unsigned gb = red_blocks(Dl);
    const int width = SQD_MAX_SPACE + 4;
    SplitRows split{nullptr, nullptr, c->nb};
    const int32_t* ri = nullptr;
    const double* pp = nullptr;
    if (shard_split_rows(c, &ri, &pp)) {
      split.rowinfo = ri;
      split.partial = pp;
    }
    const DavParams prm = shard_params(c);
    if (c->shard_max_space <= 12)
      hipLaunchKernelGGL((k_dots_eig<13>), dim3(gb), dim3(RED_T), 0, c->stream, Dl, (const double*)c->X.as<double>(),
                         c->AX.as<double>(), Dl, c->partial.as<double>(), width, counter_ptr(c), state_ptr_dev(c), prm, split);
    else
      hipLaunchKernelGGL((k_dots_eig<MAXB>), dim3(gb), dim3(RED_T), 0, c->stream, Dl, (const double*)c->X.as<double>(),
                         c->AX.as<double>(), Dl, c->partial.as<double>(), width, counter_ptr(c), state_ptr_dev(c), prm, split);
    SQD_HIP_CHECK(hipGetLastError());
  }
  double* tot2 = nullptr;
  int n2 = 0;
  SQD_TRY(shard_dav_residual(c, &tot2, &n2, /*eig_done=*/true));
  return shard_dav_orth(c, seq_out);
}
// the progress record of the iteration whose orth stage returned `seq` (or of a later one: the stop flag only rises)
int shard_dav_wait(sqd_ctx* c, long long seq, int* stopped, double* e, double* rnorm2, int* m_cur) {
  SQD_TRY(shard_check(c));
  SQD_TRY(wait_mail(c, 1, seq));
  const double* h_prog = c->h_mail + MAIL_SLOT;
  if (stopped) *stopped = h_prog[MAIL_PAYLOAD + 1] != 0.0;
  if (e) *e = h_prog[MAIL_PAYLOAD + 2];
  if (rnorm2) *rnorm2 = h_prog[MAIL_PAYLOAD + 4];
  if (m_cur) *m_cur = (int)h_prog[MAIL_PAYLOAD + 5];
  return SQD_OK;
}
int shard_dav_end(sqd_ctx* c, double** d_solution_rows, sqd_davidson_stats* st) {
  SQD_TRY(shard_check(c));
  const int64_t Dl = c->shard_Dl;
  hipLaunchKernelGGL(k_solution, dim3(red_blocks(Dl)), dim3(RED_T), 0, c->stream, Dl, (const double*)c->X.as<double>(), Dl,
                     (const DavState*)state_ptr_dev(c), c->sol.as<double>(), c->d_mail + 2 * MAIL_SLOT + MAIL_PAYLOAD, 0);
  SQD_HIP_CHECK(hipGetLastError());
  SQD_STREAM_SYNC(c->stream);
  c->shard_active = false;
  c->have_solution = true;
  c->dav_timed = false;
  c->dav_nev = 0;
  if (d_solution_rows) *d_solution_rows = c->sol.as<double>();
  return davidson_collect(c, st);
}

// outcome and event timings of the latest run (the stream must have been synchronised since)
int davidson_collect(sqd_ctx* c, sqd_davidson_stats* st) {
  const double* res = c->h_mail + 2 * MAIL_SLOT + MAIL_PAYLOAD;
  std::atomic_thread_fence(std::memory_order_acquire);
  const int iterations = (int)res[1];
  if (res[5] != 0.0) {
    set_error("initial vector has zero norm");
    return SQD_ERR_INVALID;
  }
  float ms = 0.f;
  if (c->dav_timed) SQD_HIP_CHECK(hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
  if (c->ms_setup < 0.0) {
    float tms = 0.f;
    SQD_HIP_CHECK(hipEventElapsedTime(&tms, c->ev[0], c->ev[1]));
    c->ms_setup = tms;
  }
  if (st) {
    st->converged = (int)res[0];
    st->iterations = iterations;
    st->n_sigma = (int)res[2];
    st->e_davidson = res[3];
    st->residual = std::sqrt(res[4] > 0.0 ? res[4] : 0.0);
    st->n_eig_solves = (int)res[7];
    st->n_eig_fallbacks = (int)res[8];
    st->ms_total = ms;
    double msig = 0.0, mker = 0.0, mempty = 0.0;
    int counted = 0;
    for (int i = 0; i < c->dav_nev; ++i) {
      if (c->dav_ev_iter[i] >= iterations) continue;  // enqueued ahead of the stop: that launch returned at once
      float t = 0.f;
      SQD_HIP_CHECK(hipEventElapsedTime(&t, c->sig_ev[4 * i], c->sig_ev[4 * i + 2]));
      msig += t;
      SQD_HIP_CHECK(hipEventElapsedTime(&t, c->sig_ev[4 * i], c->sig_ev[4 * i + 1]));
      mker += t;
      SQD_HIP_CHECK(hipEventElapsedTime(&t, c->sig_ev[4 * i + 2], c->sig_ev[4 * i + 3]));
      mempty += t;
      ++counted;
    }
    st->ms_sigma = msig;
    st->ms_sigma_kernel = mker;
    st->ms_event_overhead = mempty;
    st->n_sigma_timed = counted;
    st->ms_setup = c->ms_setup;
  }
  return SQD_OK;
}

}  // namespace sqd

#ifdef SQD_PHASE_CLOCK
extern "C" __attribute__((visibility("default"))) int sqd_probe_clk(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(sqd::sqd_clk), sizeof(unsigned long long) * (64 + 1024)) != hipSuccess) return -1;
  if (reset) {
    unsigned long long z[64] = {};
    z[63] = ~0ull;
    if (hipMemcpyToSymbol(HIP_SYMBOL(sqd::sqd_clk), z, sizeof z) != hipSuccess) return -1;
  }
  return 0;
}
#endif
