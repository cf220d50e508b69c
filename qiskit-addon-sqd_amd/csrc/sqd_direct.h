// The element-gather sigma kernel's arguments and its per-element body, shared with k_observables (sqd_rdm.hip), which
// evaluates S^2 c element by element in its own pass for ultra-sparse sets instead of a sigma launch of its own.
#pragma once
#include "sqd_common.h"

namespace sqd {

struct DirectArgs {
  GPtr<const double> c;
  GPtr<double> sigma;
  GPtr<const double> hdiag;
  int64_t row0, row1, nb;
  int nnorb, mode, spin;
  double ss, shift, szterm;
  GPtr<const uint64_t> strs_a, strs_b;
  GPtr<const int64_t> sa_ptr, da_ptr, sb_ptr, db_ptr;
  GPtr<const SRec> sa_rec, sb_rec;
  GPtr<const double> sa_val, sb_val;
  GPtr<const uint32_t> da_src, db_src;
  GPtr<const double> da_val, db_val;
  GPtr<const double> ja_row, jbT, eri_pp;
  GPtr<const int> stop;
  GPtr<const int> vec_index;
  int64_t c_stride, s_stride;
  // k_sigma_rows only: the beta doubles in per-slice jagged-diagonal order (k_tables_jds) and the LDS row pitch
  GPtr<const uint32_t> jd_src;
  GPtr<const double> jd_val;
  int64_t nb_pad;
  unsigned gx;  // workgroups of THIS subspace (batched launches: gridDim.x is the largest of the class)
};

// element i (relative to row0 * nb) of the operator the arguments describe, applied to C
template <bool SPIN>
__device__ inline double direct_element(const DirectArgs& g, const double* __restrict__ C, int64_t i, double pen) {
  const int64_t nb = g.nb;
  const int64_t Ar = i / nb, B = i - Ar * nb, A = g.row0 + Ar;
  const double* crow = C + A * nb;
  double a;
  if (g.mode == 0) {
    double d = g.hdiag[i];
    if (SPIN) d += g.shift * (g.szterm + (double)__popcll(g.strs_b[B] & ~g.strs_a[A]) - g.ss);
    a = d * crow[B];
  } else {
    a = (g.szterm + (double)__popcll(g.strs_b[B] & ~g.strs_a[A])) * crow[B];
  }
  const int64_t sb0 = g.sb_ptr[B], sb1 = g.sb_ptr[B + 1], sa0 = g.sa_ptr[A], sa1 = g.sa_ptr[A + 1];
  if (g.mode == 0) {
    // beta singles: same-spin value + alpha occupation term; beta doubles
    for (int64_t l = sb0; l < sb1; ++l) {
      const SRec r = g.sb_rec[l];
      a += (g.sb_val[l] + srec_sign(r.meta) * g.ja_row[A * g.nnorb + (srec_widx(r.meta) >> 1)]) * crow[r.src];
    }
    for (int64_t l = g.db_ptr[B]; l < g.db_ptr[B + 1]; ++l) a += g.db_val[l] * crow[g.db_src[l]];
    // alpha same-spin links (singles' one-body part, then doubles: the CSR lists as the table build left them,
    // in the order of the merged list the work-item kernel reads), then alpha singles x beta occupation
    for (int64_t l = sa0; l < sa1; ++l) a += g.sa_val[l] * C[(int64_t)g.sa_rec[l].src * nb + B];
    for (int64_t l = g.da_ptr[A]; l < g.da_ptr[A + 1]; ++l) a += g.da_val[l] * C[(int64_t)g.da_src[l] * nb + B];
    for (int64_t l = sa0; l < sa1; ++l) {
      const SRec r = g.sa_rec[l];
      a += srec_sign(r.meta) * g.jbT[(int64_t)(srec_widx(r.meta) >> 1) * nb + B] * C[(int64_t)r.src * nb + B];
    }
  }
  // single x single (and the S^2 exchange term: the beta link that undoes the alpha link's orbital move)
  for (int64_t la = sa0; la < sa1; ++la) {
    const SRec ra = g.sa_rec[la];
    const double* srow = C + (int64_t)ra.src * nb;
    const double* w = g.eri_pp + (int64_t)(srec_widx(ra.meta) >> 1) * g.nnorb;
    const int partner = (int)srec_widx(ra.meta) ^ 1;
    double t = 0.0;
    for (int64_t lb = sb0; lb < sb1; ++lb) {
      const SRec rb = g.sb_rec[lb];
      double wv = (g.mode == 0) ? w[srec_widx(rb.meta) >> 1] : 0.0;
      if (SPIN) wv += ((int)srec_widx(rb.meta) == partner) ? pen : 0.0;
      t += srec_sign(rb.meta) * wv * srow[rb.src];
    }
    a += srec_sign(ra.meta) * t;
  }
  return a;
}

// fills the arguments for the current subspace (sqd_sigma.hip); d_sigma may be nullptr when only direct_element is used
void fill_direct_args(sqd_ctx* c, const double* d_c, double* d_sigma, int mode, bool spin, double ss, double shift,
                      int64_t in_stride, int64_t out_stride, DirectArgs* g);

}  // namespace sqd
