// The Davidson state block and the start vector, shared by sqd_davidson.hip (the solver) and sqd_tables.hip (whose
// last table-build launch also prepares the run that normally follows: state block + pyscf's start vector).
#pragma once
#include "sqd_common.h"
#include "sqd_device.h"

namespace sqd {

// ------------------------------------------------------------------ the state block
constexpr int MAXB = SQD_MAX_SPACE + 1;  // most basis vectors a run can hold (max_space + the fresh correction)

struct DavState {
  int m_next;    // size of the basis whose newest vector X[m_next-1] is next to get its sigma (read by k_sigma, k_dots_eig)
  int m_cur;     // size of the basis of the CURRENT projected problem (written by k_dots_eig; read by residual / orth)
  int it;        // projected problems solved so far
  int nsig;      // sigma builds that entered the projected matrix
  int stop;      // != 0: the solve is over, every kernel enqueued behind this returns at once.  1: raised by the eigen step;
                 // >= 2: by a k_orth_dev launch, the value is that launch's mark (its sequence number folded to an int) -- a
                 // workgroup of THAT launch which starts late and finds the flag up still has its share of the solution to
                 // form (single solves: sol_out); ONE word, so no ordering between a flag and a mark is needed
  int conv;      // ... and converged
  int first;     // no projected problem solved yet (dE of the first one is the eigenvalue itself)
  int m_eig;     // size of the last projected problem solved (-1: none, e.g. right after a restart)
  int restart;   // the current iteration collapses the basis to {Ritz vector, correction}
  int err;       // 1: the start vector has zero norm
  int sol_m;     // the solution is sum_{v < sol_m} sol_coef[v] X_v
  int n_rqi;     // diagnostics: shifted solves of the warm-started eigen-solver, and how often it gave way to Jacobi
  int n_jacobi;
  int pad;
  double e, de, rnorm2;
  double sv[MAXB + 1];        // 1 / |X_v| (basis vector v is sv_v X_v: vectors are never normalised by a pass)
  double coef[MAXB + 1];      // Ritz coefficients on the orthonormal basis
  double raw[MAXB + 1];       // ... on the stored vectors (coef * sv)
  double sol_coef[MAXB + 1];  // normalised raw coefficients of the solution
  double heff[MAXB * MAXB];   // projected matrix, row stride MAXB
};
struct DavParams {  // constants of one run
  double tol, tol2, lindep;
  int max_space;
};

// start of a run: state block and arrival counters (workgroup 0; the kernels that use them come later in the stream)
__device__ inline void dav_state_init(DavState* st, unsigned* counter) {
  for (int i = threadIdx.x; i < COUNT_WORDS; i += blockDim.x) counter[i] = 0u;
  for (int i = threadIdx.x; i < MAXB * MAXB; i += blockDim.x) st->heff[i] = 0.0;
  for (int i = threadIdx.x; i <= MAXB; i += blockDim.x) {
    st->sv[i] = (i == 0) ? 1.0 : 0.0;
    st->coef[i] = (i == 0) ? 1.0 : 0.0;
    st->raw[i] = (i == 0) ? 1.0 : 0.0;
    st->sol_coef[i] = (i == 0) ? 1.0 : 0.0;
  }
  if (threadIdx.x == 0) {
    st->m_next = 1;
    st->m_cur = 1;
    st->it = 0;
    st->nsig = 0;
    st->stop = 0;
    st->conv = 0;
    st->first = 1;
    st->m_eig = -1;
    st->restart = 0;
    st->err = 0;
    st->sol_m = 1;
    st->n_rqi = 0;
    st->n_jacobi = 0;
    st->e = 0.0;
    st->de = 0.0;
    st->rnorm2 = 0.0;
  }
}

// pyscf get_init_guess: unit vector at addr, +1e-5 on the first and -1e-5 on the last element,
// normalised here with the closed-form norm (no reduction, no host round trip).
// Every workgroup repeats the final stage of the argmin over the per-row candidates (pmin / pidx, left by
// k_tables_diag) instead of a separate single-workgroup launch; thread gtid of gthreads writes its share of x.
// All threads of the workgroup must call this (block reduction inside).
__device__ inline void init_guess_write(int64_t n, const double* __restrict__ pmin, const int64_t* __restrict__ pidx,
                                        int nblocks, double* __restrict__ x, int64_t gtid, int64_t gthreads) {
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
  for (int64_t i = gtid; i < n; i += gthreads) x[i] = f(i) * inv;
}

}  // namespace sqd
