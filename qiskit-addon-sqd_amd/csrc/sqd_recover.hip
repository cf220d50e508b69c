// Host-side half of configuration recovery (no device code in this file).
//
// Reference qiskit_addon_sqd/configuration_recovery.py:230-304 repairs one bitstring at a time in Python:
// per half (spin-down = left, spin-up = right) it draws the bits to flip with
//     rng.choice(candidates, size=|excess|, replace=False, p=weights)
// Once the solver takes ~1 ms per batch this loop is the wall-clock of an SQD iteration (measured: 1.5 s
// per iteration at 1e5 samples against 30 ms for the eight solves).  A seeded run must keep numpy's random
// stream, so the selection cannot be re-designed -- but it can be replayed: numpy's Generator.choice with
// p and replace=False consumes plain uniform doubles in a fixed pattern
//     while n_found < size:  x = random(size - n_found); p[found] = 0; cdf = cumsum(p) / cdf[-1];
//                            new = searchsorted(cdf, x, 'right'), first occurrences kept in order
// (numpy/random/_generator.pyx).  The Python side draws a block of uniforms (an exact upper bound on what
// the rows can consume), this routine replays the selections over all rows, reports how many doubles it
// used, and the caller rewinds the PCG64 stream by the rest (BitGenerator.advance).  Floating point is
// replayed operation by operation, including numpy's pairwise summation, so that every cdf is bit-identical
// to the reference's; tests/test_sqd_loop.py replays a recorded run of the reference against it.
#include <cmath>
#include <cstdint>

#include "sqd_common.h"

namespace sqd {

// numpy's float64 add.reduce over a contiguous 1-D array (loops_utils.h: DOUBLE_pairwise_sum)
static double np_pairwise_sum(const double* a, int n) {
  if (n < 8) {
    double res = 0.0;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

struct UniformStream {
  const double* u;
  int64_t n, pos;
};

// One half of one row.  Returns 0, 1 (stream exhausted) or 2 (a condition numpy raises on: let Python do it).
static int repair_half(uint8_t* b, int n, const double* w_up, const double* w_down, int target, UniformStream& us) {
  double w[SQD_MAX_NORB], sub[SQD_MAX_NORB], p[SQD_MAX_NORB], cdf[SQD_MAX_NORB];
  int cand[SQD_MAX_NORB], found[SQD_MAX_NORB], fresh[SQD_MAX_NORB];
  bool any = false;
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    double x = b[i] ? w_down[i] : w_up[i];
    x = std::fmax(0.0, x);  // np.maximum / np.minimum propagate NaN; fmax does not: checked below
    x = std::fmin(1.0, x);
    if (std::isnan(w_down[i]) || std::isnan(w_up[i])) return 2;
    w[i] = x;
    any = any || (x != 0.0);
    cnt += b[i] ? 1 : 0;
  }
  if (!any) return 0;
  const double s = np_pairwise_sum(w, n);
  for (int i = 0; i < n; ++i) w[i] /= s;
  const int excess = cnt - target;
  if (excess == 0) return 0;
  const uint8_t want = excess > 0 ? 1 : 0;  // flip occupied bits down, or empty bits up
  const int size = excess > 0 ? excess : -excess;
  int m = 0;
  for (int i = 0; i < n; ++i)
    if ((b[i] != 0) == (want != 0)) {
      cand[m] = i;
      sub[m] = w[i];
      ++m;
    }
  const double ps = np_pairwise_sum(sub, m);
  int nonzero = 0;
  for (int k = 0; k < m; ++k) {
    p[k] = sub[k] / ps;
    if (std::isnan(p[k])) return 2;
    nonzero += p[k] > 0.0 ? 1 : 0;
  }
  if (size > m || nonzero < size) return 2;
  int nf = 0;
  while (nf < size) {
    const int k = size - nf;
    if (us.pos + k > us.n) return 1;
    const double* x = us.u + us.pos;
    us.pos += k;
    for (int f = 0; f < nf; ++f) p[found[f]] = 0.0;
    double run = 0.0;
    for (int i = 0; i < m; ++i) {
      run += p[i];
      cdf[i] = run;
    }
    const double last = cdf[m - 1];
    for (int i = 0; i < m; ++i) cdf[i] /= last;
    // searchsorted(side='right') per draw, then the first occurrence of every index, in draw order
    int nfresh = 0;
    for (int j = 0; j < k; ++j) {
      int lo = 0, hi = m;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= x[j]) lo = mid + 1;
        else hi = mid;
      }
      bool seen = false;
      for (int q = 0; q < nfresh; ++q) seen = seen || (fresh[q] == lo);
      if (!seen) fresh[nfresh++] = lo;
    }
    for (int q = 0; q < nfresh; ++q) {
      if (fresh[q] >= m) return 2;  // cannot happen for x < 1; numpy would raise an IndexError
      found[nf++] = fresh[q];
    }
  }
  for (int f = 0; f < size; ++f) b[cand[found[f]]] = want ? 0 : 1;
  return 0;
}

}  // namespace sqd

using namespace sqd;

extern "C" __attribute__((visibility("default"))) int sqd_recover_rows(
    uint8_t* bits, int64_t n_total, int norb, const int64_t* rows, int64_t nrows, const double* up_left,
    const double* down_left, const double* up_right, const double* down_right, int target_left, int target_right,
    const double* uniforms, int64_t n_uniforms, int64_t* n_used) {
  if (!bits || !rows || !uniforms || !n_used || norb < 1 || norb > SQD_MAX_NORB) {
    set_error("sqd_recover_rows: bad argument");
    return SQD_ERR_INVALID;
  }
  UniformStream us{uniforms, n_uniforms, 0};
  for (int64_t r = 0; r < nrows; ++r) {
    const int64_t i = rows[r];
    if (i < 0 || i >= n_total) {
      set_error("sqd_recover_rows: row index out of range");
      return SQD_ERR_INVALID;
    }
    uint8_t* row = bits + i * 2 * (int64_t)norb;
    // left (spin-down) half first, then right (spin-up): the reference's stream order
    int rc = repair_half(row, norb, up_left, down_left, target_left, us);
    if (rc == 0) rc = repair_half(row + norb, norb, up_right, down_right, target_right, us);
    if (rc == 1) {
      set_error("sqd_recover_rows: uniform stream exhausted");
      return SQD_ERR_LIMIT;
    }
    if (rc == 2) {
      set_error("sqd_recover_rows: weights that numpy's Generator.choice rejects (row " + std::to_string(i) + ")");
      return SQD_ERR_STATE;
    }
  }
  *n_used = us.pos;
  return SQD_OK;
}
