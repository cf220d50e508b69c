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
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

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
  // (nothing to repair in this half: no weights are formed and no random number is drawn -- the reference returns from
  // the same two conditions, in the other order)
  int cnt = 0;
  for (int i = 0; i < n; ++i) cnt += b[i] ? 1 : 0;
  const int excess = cnt - target;
  if (excess == 0) return 0;
  bool any = false;
  for (int i = 0; i < n; ++i) {
    const double x = b[i] ? w_down[i] : w_up[i];  // (clipped to [0, 1] and checked for NaN once per call)
    w[i] = x;
    any = any || (x != 0.0);
  }
  if (!any) return 0;
  const double s = np_pairwise_sum(w, n);
  for (int i = 0; i < n; ++i) w[i] /= s;
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

// ---- The same selection without a single division (round 5).  The draw only needs to know WHERE x falls among the
// cumulative probabilities, and the normalisations cancel: cdf[i] = (sum_{k<=i} w_k) / (sum_k w_k) up to rounding, whatever
// `s` and `ps` were.  Every cdf[i] the reference forms is within (2m + 22) * 2^-53 < 2e-14 (m <= 64 candidates) of that
// real number, and so is the plain running sum R[i] / R[m-1] of the raw weights.  So x is located among R[i] against
// x * R[m-1], and unless it lies within 1e-11 (relative to the total) of a boundary -- 500x the two error bounds together
// -- the index is the reference's.  Anything else (a near tie, bytes that are not 0 / 1, inputs numpy raises on) returns
// -1 with nothing consumed or changed, and the operation-by-operation replay above decides.  A half costs its popcount,
// m additions and a binary search per draw instead of 30 + 3m divisions, two pairwise sums and a branchy compaction.
static inline bool half_mask(const uint8_t* b, int n, uint64_t* out) {
  uint64_t m = 0, odd = 0;
  int i = 0;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    std::memcpy(&w, b + i, 8);
    odd |= w & 0xFEFEFEFEFEFEFEFEull;
    // eight 0 / 1 bytes -> eight bits with one multiplication (byte j lands on bit 56 + j of the product)
    m |= (((w & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56) << i;
  }
  for (; i < n; ++i) {
    odd |= b[i] & 0xFEu;
    m |= (uint64_t)(b[i] & 1) << i;
  }
  *out = m;
  return odd == 0;
}

struct FlipWeights {  // one half's clipped weights and which of them are nonzero
  const double *up, *down;
  uint64_t nz_up, nz_down;
};

static int repair_half_fast(uint8_t* b, int n, const FlipWeights& fw, int target, UniformStream& us) {
  uint64_t mask;
  if (!half_mask(b, n, &mask)) return -1;
  const uint64_t full = n >= 64 ? ~0ull : ((1ull << n) - 1);
  const int excess = __builtin_popcountll(mask) - target;
  if (excess == 0) return 0;
  if (((mask & fw.nz_down) | (~mask & full & fw.nz_up)) == 0) return 0;  // every flip weight zero: nothing is drawn
  const bool down = excess > 0;  // flip occupied bits down, or empty bits up
  const int size = down ? excess : -excess;
  const uint64_t cmask = down ? mask : (~mask & full);
  const double* wt = down ? fw.down : fw.up;
  const int m = __builtin_popcountll(cmask);
  if (size > m || __builtin_popcountll(cmask & (down ? fw.nz_down : fw.nz_up)) < size) return -1;
  double wv[SQD_MAX_NORB], R[SQD_MAX_NORB];
  int cand[SQD_MAX_NORB], found[SQD_MAX_NORB], fresh[SQD_MAX_NORB];
  {
    int k = 0;
    for (uint64_t t = cmask; t; t &= t - 1, ++k) {
      cand[k] = __builtin_ctzll(t);
      wv[k] = wt[cand[k]];
    }
  }
  const int64_t pos0 = us.pos;
  int nf = 0;
  while (nf < size) {
    const int k = size - nf;
    if (us.pos + k > us.n) return 1;
    const double* x = us.u + us.pos;
    us.pos += k;
    for (int f = 0; f < nf; ++f) wv[found[f]] = 0.0;
    double acc = 0.0;
    for (int i = 0; i < m; ++i) {
      acc += wv[i];
      R[i] = acc;
    }
    const double W = R[m - 1];
    if (!(W > 0.0)) {
      us.pos = pos0;
      return -1;
    }
    const double tol = W * 1e-11;
    int nfresh = 0;
    for (int j = 0; j < k; ++j) {
      const double t = x[j] * W;
      int lo = 0, hi = m;  // lo = #{i : R[i] <= t}
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (R[mid] <= t) lo = mid + 1;
        else hi = mid;
      }
      if (lo >= m || R[lo] - t <= tol || (lo > 0 && t - R[lo - 1] <= tol)) {
        us.pos = pos0;
        return -1;
      }
      bool seen = false;
      for (int q = 0; q < nfresh; ++q) seen = seen || (fresh[q] == lo);
      if (!seen) fresh[nfresh++] = lo;
    }
    for (int q = 0; q < nfresh; ++q) found[nf++] = fresh[q];
  }
  for (int f = 0; f < size; ++f) b[cand[found[f]]] = down ? 0 : 1;
  return 0;
}

}  // namespace sqd

using namespace sqd;

extern "C" __attribute__((visibility("default"))) int sqd_recover_rows(
    uint8_t* bits, int64_t n_total, int norb, const int64_t* rows, int64_t nrows, const double* up_left,
    const double* down_left, const double* up_right, const double* down_right, int target_left, int target_right,
    const double* uniforms, int64_t n_uniforms, int64_t* n_used) {
  if (!bits || !uniforms || !n_used || norb < 1 || norb > SQD_MAX_NORB) {  // (rows == NULL: every row, in order)
    set_error("sqd_recover_rows: bad argument");
    return SQD_ERR_INVALID;
  }
  UniformStream us{uniforms, n_uniforms, 0};
  // np.minimum(1, np.maximum(0, weights)) once per call instead of once per half-row; NaN weights: Python's job
  double cl[4][SQD_MAX_NORB];
  const double* src[4] = {up_left, down_left, up_right, down_right};
  for (int t = 0; t < 4; ++t)
    for (int i = 0; i < norb; ++i) {
      if (std::isnan(src[t][i])) {
        set_error("sqd_recover_rows: NaN weight");
        return SQD_ERR_STATE;
      }
      cl[t][i] = std::fmin(1.0, std::fmax(0.0, src[t][i]));
    }
  up_left = cl[0], down_left = cl[1], up_right = cl[2], down_right = cl[3];
  FlipWeights fw[2] = {{up_left, down_left, 0, 0}, {up_right, down_right, 0, 0}};
  for (int i = 0; i < norb; ++i) {
    fw[0].nz_up |= (uint64_t)(up_left[i] != 0.0) << i, fw[0].nz_down |= (uint64_t)(down_left[i] != 0.0) << i;
    fw[1].nz_up |= (uint64_t)(up_right[i] != 0.0) << i, fw[1].nz_down |= (uint64_t)(down_right[i] != 0.0) << i;
  }
  static const bool exact_only = std::getenv("SQD_RECOVER_EXACT") != nullptr;  // (test hook: the replay alone)
  for (int64_t r = 0; r < nrows; ++r) {
    const int64_t i = rows ? rows[r] : r;
    if (i < 0 || i >= n_total) {
      set_error("sqd_recover_rows: row index out of range");
      return SQD_ERR_INVALID;
    }
    uint8_t* row = bits + i * 2 * (int64_t)norb;
    // left (spin-down) half first, then right (spin-up): the reference's stream order
    int rc = exact_only ? -1 : repair_half_fast(row, norb, fw[0], target_left, us);
    if (rc < 0) rc = repair_half(row, norb, up_left, down_left, target_left, us);
    if (rc == 0) {
      rc = exact_only ? -1 : repair_half_fast(row + norb, norb, fw[1], target_right, us);
      if (rc < 0) rc = repair_half(row + norb, norb, up_right, down_right, target_right, us);
    }
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

// ---- the rest of the host side of an SQD iteration's sample processing, natively: at 1e5 samples the numpy
// passes around the repair -- Hamming weights of every row, the duplicate merge, the subsampling draws -- cost more than
// the repair itself and, together, several times the batched solve they feed.
//
// sqd_hamming_excess: per row the distance of both halves from their target weights; *bound = an upper bound on the
// uniform doubles sqd_recover_rows can consume (a draw of k bits takes k, then at most k - 1, ... doubles), *nbad = rows
// off target.  bits: [n][2 norb] bytes (0 / 1).
extern "C" __attribute__((visibility("default"))) int sqd_hamming_excess(const uint8_t* bits, int64_t n, int norb,
                                                                          int target_left, int target_right,
                                                                          int64_t* bound, int64_t* nbad) {
  if (!bits || !bound || !nbad || norb < 1) return SQD_ERR_INVALID;
  int64_t bd = 0, bad = 0;
  for (int64_t r = 0; r < n; ++r) {
    const uint8_t* row = bits + r * 2 * (int64_t)norb;
    int cl = 0, cr = 0;
    for (int i = 0; i < norb; ++i) cl += row[i], cr += row[norb + i];
    const int64_t el = std::abs(cl - target_left), er = std::abs(cr - target_right);
    bd += el * (el + 1) / 2 + er * (er + 1) / 2;
    bad += (el || er) ? 1 : 0;
  }
  *bound = bd;
  *nbad = bad;
  return SQD_OK;
}

// sqd_merge_rows: duplicates of a bool matrix merged in FIRST-OCCURRENCE order, their probabilities added one by one in
// row order (what the reference's running dictionary does, configuration_recovery.py:112-126).  first[k] = row index of
// the k-th distinct row, freq[k] = its summed probability (not normalised); *n_unique.  nbits <= 128.  With compact != 0
// the distinct rows are also moved to the front of `bits` in that order (first[k] >= k: in place), which saves the caller a
// gather.  Keys and home slots are formed in a first pass so that the insertion can prefetch the slot of the row eight
// ahead (the table of 1e5 rows lives in L2: every probe is a dependent miss otherwise).
extern "C" __attribute__((visibility("default"))) int sqd_merge_rows(uint8_t* bits, int64_t n, int nbits,
                                                                      const double* probs, int64_t* first, double* freq,
                                                                      int64_t* n_unique, int compact) {
  if (!bits || !probs || !first || !freq || !n_unique || nbits < 1 || nbits > 128 || n < 0) return SQD_ERR_INVALID;
  if (n > 0x3fffffff) return SQD_ERR_LIMIT;
  size_t cap = 16;
  while (cap < (size_t)n * 2 + 16) cap <<= 1;
  std::vector<int32_t> slot(cap, -1);  // -> index into first / freq / ukeys
  std::vector<uint64_t> keys((size_t)n * 2), ukeys((size_t)n * 2);  // (k0, k1) per row / per distinct row
  std::vector<uint32_t> home((size_t)n);
  for (int64_t r = 0; r < n; ++r) {
    const uint8_t* row = bits + r * (int64_t)nbits;
    // eight 0 / 1 bytes -> eight bits with one multiplication (byte j lands on bit 56 + j of the product)
    uint64_t k0 = 0, k1 = 0;
    int i = 0;
    for (; i + 8 <= nbits; i += 8) {
      uint64_t w;
      std::memcpy(&w, row + i, 8);
      const uint64_t b8 = ((w & 0x0101010101010101ull) * 0x0102040810204080ull) >> 56;
      if (i < 64) k0 |= b8 << i;
      else k1 |= b8 << (i - 64);
    }
    for (; i < nbits; ++i) {
      if (i < 64) k0 |= (uint64_t)(row[i] & 1) << i;
      else k1 |= (uint64_t)(row[i] & 1) << (i - 64);
    }
    uint64_t h = (k0 ^ (k1 * 0x9E3779B97F4A7C15ull)) * 0xD6E8FEB86659FD93ull;
    h ^= h >> 32;
    keys[2 * (size_t)r] = k0;
    keys[2 * (size_t)r + 1] = k1;
    home[(size_t)r] = (uint32_t)((size_t)h & (cap - 1));
  }
  int64_t nu = 0;
  for (int64_t r = 0; r < n; ++r) {
    if (r + 8 < n) __builtin_prefetch(&slot[home[(size_t)r + 8]]);
    const uint64_t k0 = keys[2 * (size_t)r], k1 = keys[2 * (size_t)r + 1];
    size_t pos = home[(size_t)r];
    for (;;) {
      const int32_t e = slot[pos];
      if (e < 0) {
        slot[pos] = (int32_t)nu;
        ukeys[2 * (size_t)nu] = k0;
        ukeys[2 * (size_t)nu + 1] = k1;
        first[nu] = r;
        freq[nu] = probs[r];
        ++nu;
        break;
      }
      if (ukeys[2 * (size_t)e] == k0 && ukeys[2 * (size_t)e + 1] == k1) {
        freq[e] += probs[r];
        break;
      }
      pos = (pos + 1) & (cap - 1);
    }
  }
  if (compact)
    for (int64_t k = 0; k < nu; ++k)
      if (first[k] != k) std::memcpy(bits + k * (int64_t)nbits, bits + first[k] * (int64_t)nbits, (size_t)nbits);
  *n_unique = nu;
  return SQD_OK;
}

// sqd_unique_rows: np.unique(bool_matrix, axis=0, return_counts=True) of counts.py:45-61 (the 1e5 shots of a BitArray ->
// distinct bitstrings in lexicographic row order + how often each occurred): rows packed MSB-first into one or two 64-bit
// keys, a stable LSD radix sort of (key, row index) on 16-bit digits (digits every row agrees on are skipped), runs
// counted.  first[k] = smallest row index of the k-th distinct row, counts[k] its multiplicity.  nbits <= 128; bytes
// other than 0 / 1 -> SQD_ERR_STATE (the caller then asks numpy).
extern "C" __attribute__((visibility("default"))) int sqd_unique_rows(const uint8_t* bits, int64_t n, int nbits,
                                                                       int64_t* first, int64_t* counts,
                                                                       int64_t* n_unique) {
  if (!bits || !first || !counts || !n_unique || nbits < 1 || n < 0) return SQD_ERR_INVALID;
  if (nbits > 128 || n > 0x7fffffff) return SQD_ERR_LIMIT;
  const int words = nbits > 64 ? 2 : 1;
  std::vector<uint64_t> key[2] = {std::vector<uint64_t>((size_t)n * words), std::vector<uint64_t>((size_t)n * words)};
  std::vector<uint32_t> idx[2] = {std::vector<uint32_t>((size_t)n), std::vector<uint32_t>((size_t)n)};
  uint64_t odd = 0;
  for (int64_t r = 0; r < n; ++r) {
    const uint8_t* row = bits + r * (int64_t)nbits;
    uint64_t k[2] = {0, 0};  // column c -> bit 63 - (c mod 64) of word c / 64: integer order = row order
    int c = 0;
    for (; c + 8 <= nbits; c += 8) {
      uint64_t w;
      std::memcpy(&w, row + c, 8);
      odd |= w & 0xFEFEFEFEFEFEFEFEull;
      // eight 0 / 1 bytes -> one byte, FIRST column in the top bit (byte j lands on bit 63 - j of the product)
      k[c >> 6] |= (((w & 0x0101010101010101ull) * 0x8040201008040201ull) >> 56) << (56 - (c & 63));
    }
    for (; c < nbits; ++c) {
      odd |= row[c] & 0xFEu;
      k[c >> 6] |= (uint64_t)(row[c] & 1) << (63 - (c & 63));
    }
    for (int w = 0; w < words; ++w) key[0][(size_t)r * words + w] = k[w];
    idx[0][(size_t)r] = (uint32_t)r;
  }
  if (odd) return SQD_ERR_STATE;
  int cur = 0;
  std::vector<uint32_t> hist(65536);
  for (int w = words - 1; w >= 0; --w)  // least significant word first
    for (int shift = 0; shift < 64; shift += 16) {
      std::fill(hist.begin(), hist.end(), 0u);
      const uint64_t* kc = key[cur].data();
      for (int64_t r = 0; r < n; ++r) ++hist[(kc[(size_t)r * words + w] >> shift) & 0xFFFF];
      if (n && hist[(kc[(size_t)w] >> shift) & 0xFFFF] == (uint32_t)n) continue;  // every row agrees on this digit
      uint32_t run = 0;
      for (auto& h : hist) {
        const uint32_t c0 = h;
        h = run;
        run += c0;
      }
      uint64_t* kn = key[cur ^ 1].data();
      const uint32_t* ic = idx[cur].data();
      uint32_t* in = idx[cur ^ 1].data();
      for (int64_t r = 0; r < n; ++r) {
        const uint32_t to = hist[(kc[(size_t)r * words + w] >> shift) & 0xFFFF]++;
        for (int q = 0; q < words; ++q) kn[(size_t)to * words + q] = kc[(size_t)r * words + q];
        in[to] = ic[r];
      }
      cur ^= 1;
    }
  const uint64_t* ks = key[cur].data();
  const uint32_t* is = idx[cur].data();
  int64_t nu = 0;
  for (int64_t r = 0; r < n; ++r) {
    bool same = r > 0;
    for (int q = 0; q < words && same; ++q) same = ks[(size_t)r * words + q] == ks[(size_t)(r - 1) * words + q];
    if (same) ++counts[nu - 1];
    else first[nu] = is[r], counts[nu] = 1, ++nu;  // (stable sort: the first of a run is its smallest row index)
  }
  *n_unique = nu;
  return SQD_OK;
}

// sqd_choice_replay: numpy's Generator.choice(n, size, replace=False, p=p) (subsampling.py:200-207: one call per batch)
// replayed on a block of uniforms, as sqd_recover_rows does for the repair draws: out[size] indices, *n_used doubles
// consumed (size at least; more only when two draws of a round land on the same element).  Returns SQD_ERR_STATE for
// inputs numpy raises on (the caller then makes the numpy call, which raises), SQD_ERR_LIMIT for a short stream.
extern "C" __attribute__((visibility("default"))) int sqd_choice_replay(const double* p_in, int64_t n, int64_t size,
                                                                         int64_t nbatches, const double* uniforms,
                                                                         int64_t n_uniforms, int64_t* out, int64_t* n_used) {
  if (!p_in || !uniforms || !out || !n_used || n < 1 || size < 1 || nbatches < 1) return SQD_ERR_INVALID;
  int64_t nonzero = 0;
  for (int64_t i = 0; i < n; ++i) {
    if (!(p_in[i] >= 0.0) || !std::isfinite(p_in[i])) return SQD_ERR_STATE;
    nonzero += p_in[i] > 0.0 ? 1 : 0;
  }
  if (size > n || nonzero < size) return SQD_ERR_STATE;
  {  // numpy's check that p sums to one (Kahan summation, tolerance sqrt(eps)): its ValueError is Python's to raise
    double sum = p_in[0], c = 0.0;
    for (int64_t i = 1; i < n; ++i) {
      const double y = p_in[i] - c, t = sum + y;
      c = (t - sum) - y;
      sum = t;
    }
    if (std::fabs(sum - 1.0) > 1.4901161193847656e-08) return SQD_ERR_STATE;
  }
  // the first round of every batch works on the untouched p: one cumulative sum serves them all
  std::vector<double> cdf0((size_t)n), p, cdf;
  {
    double run = 0.0;
    for (int64_t i = 0; i < n; ++i) {  // np.cumsum: a plain running sum
      run += p_in[i];
      cdf0[i] = run;
    }
    const double last = cdf0[n - 1];
    for (int64_t i = 0; i < n; ++i) cdf0[i] /= last;
  }
  std::vector<char> taken((size_t)n, 0);
  std::vector<int64_t> fresh;
  int64_t pos = 0;
  for (int64_t bt = 0; bt < nbatches; ++bt) {
    int64_t* found = out + bt * size;
    int64_t nf = 0;
    const double* cur = cdf0.data();
    while (nf < size) {
      const int64_t k = size - nf;
      if (pos + k > n_uniforms) return SQD_ERR_LIMIT;
      const double* x = uniforms + pos;
      pos += k;
      if (nf > 0) {  // a later round: the elements found so far drop out of p
        if (p.empty()) p.assign(p_in, p_in + n), cdf.resize((size_t)n);
        else if (cur == cdf0.data()) std::copy(p_in, p_in + n, p.begin());
        for (int64_t f = 0; f < nf; ++f) p[found[f]] = 0.0;
        double run = 0.0;
        for (int64_t i = 0; i < n; ++i) {
          run += p[i];
          cdf[i] = run;
        }
        const double last = cdf[n - 1];
        for (int64_t i = 0; i < n; ++i) cdf[i] /= last;
        cur = cdf.data();
      }
      fresh.clear();
      for (int64_t j = 0; j < k; ++j) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {  // searchsorted(side='right')
          const int64_t mid = (lo + hi) >> 1;
          if (cur[mid] <= x[j]) lo = mid + 1;
          else hi = mid;
        }
        if (lo >= n) return SQD_ERR_STATE;
        if (!taken[lo]) {  // first occurrence of every index, in draw order
          taken[lo] = 1;
          fresh.push_back(lo);
        }
      }
      for (int64_t q : fresh) found[nf++] = q;
    }
    for (int64_t f = 0; f < size; ++f) taken[found[f]] = 0;  // (the next batch draws from the whole set again)
  }
  *n_used = pos;
  return SQD_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Digest of the integral tensors on native threads.  A solver context is found by a hash of EVERY byte of the tensors the
// caller passes (a writeable numpy array may have been edited in place since the last call); the Python layer launches
// the solve on the previous call's context and verifies the bytes meanwhile (fermion._run_on_context).  A Python-side
// hash holds the GIL for its whole pass and runs on one core: 0.2 ms for the 6.5 MB of norb = 30 beside a 0.16 ms solve.
// Here the range is cut into fixed 512 KB pieces hashed by a small pool of native threads (four 64-bit lanes of
// multiply-rotate rounds per piece, the piece digests folded in order: the result does not depend on the number of
// threads); workers spin for a while after a job -- a solve loop hands them the next one within that time -- and sleep
// on a condition variable otherwise.
// ---------------------------------------------------------------------------------------------------------------
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <pthread.h>
#include <thread>

namespace sqd {
namespace {

constexpr uint64_t HP1 = 0x9E3779B185EBCA87ull, HP2 = 0xC2B2AE3D27D4EB4Full, HP3 = 0x165667B19E3779F9ull;
constexpr size_t HASH_PIECE = 512 * 1024;
inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
inline uint64_t hround(uint64_t acc, uint64_t in) { return rotl64(acc + in * HP2, 31) * HP1; }
inline uint64_t avalanche(uint64_t h) {
  h ^= h >> 33;
  h *= HP2;
  h ^= h >> 29;
  h *= HP3;
  h ^= h >> 32;
  return h;
}
// One 512 KB piece.  The bulk loop is the accumulate step of XXH3 (8 lanes of 64 bits per 64-byte stripe: the word is
// added to the neighbouring lane, the product of the two halves of word ^ key to its own; every 16 stripes the lanes are
// scrambled) written so that the compiler vectorises it: plain C++ compiled twice, for the baseline ISA and for AVX2
// (chosen once at run time) -- 32 x 32 -> 64-bit multiplies, no 64-bit multiplier on the critical path.
constexpr uint64_t HKEY[8] = {0xbe4ba423396cfeb8ull, 0x1cad21f72c81017cull, 0xdb979083e96dd4deull, 0x1f67b3b7a4a44072ull,
                              0x78e5c0cc4ee679cbull, 0x2172ffcc7dd05a82ull, 0x8e2443f7744608b8ull, 0x4c263a81e69035e0ull};
#define SQD_HASH_BULK_BODY                                                                   \
  uint64_t acc[8];                                                                           \
  for (int l = 0; l < 8; ++l) acc[l] = HKEY[l] ^ (seed + (uint64_t)l * HP1);                 \
  size_t i = 0, stripe = 0;                                                                  \
  for (; i + 64 <= n; i += 64, ++stripe) {                                                   \
    uint64_t w[8];                                                                           \
    std::memcpy(w, p + i, 64);                                                               \
    for (int l = 0; l < 8; ++l) {                                                            \
      const uint64_t k = w[l] ^ HKEY[l];                                                     \
      acc[l ^ 1] += w[l];                                                                    \
      acc[l] += (k & 0xffffffffull) * (k >> 32);                                             \
    }                                                                                        \
    if ((stripe & 15) == 15)                                                                 \
      for (int l = 0; l < 8; ++l) acc[l] = ((acc[l] ^ (acc[l] >> 47)) ^ HKEY[7 - l]) * 0x9E3779B1ull; \
  }                                                                                          \
  for (int l = 0; l < 8; ++l) out[l] = acc[l];                                               \
  return i;
size_t hash_bulk_base(const unsigned char* p, size_t n, uint64_t seed, uint64_t* out) { SQD_HASH_BULK_BODY }
#if defined(__x86_64__)
__attribute__((target("avx2"))) size_t hash_bulk_avx2(const unsigned char* p, size_t n, uint64_t seed, uint64_t* out) {
  SQD_HASH_BULK_BODY
}
#endif
uint64_t hash_piece(const unsigned char* p, size_t n, uint64_t seed) {
  uint64_t acc[8];
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  size_t i = avx2 ? hash_bulk_avx2(p, n, seed, acc) : hash_bulk_base(p, n, seed, acc);
#else
  size_t i = hash_bulk_base(p, n, seed, acc);
#endif
  uint64_t h = (uint64_t)n * HP1;
  for (int l = 0; l < 8; ++l) h = rotl64(h ^ avalanche(acc[l] + HKEY[l]), 23) * HP1 + HP2;
  for (; i + 8 <= n; i += 8) {
    uint64_t w;
    std::memcpy(&w, p + i, 8);
    h = rotl64(h ^ hround(0, w), 27) * HP1 + HP3;
  }
  for (; i < n; ++i) h = rotl64(h ^ (p[i] * HP3), 11) * HP1;
  return avalanche(h);
}

struct HashJob {
  const unsigned char* p[2] = {nullptr, nullptr};
  size_t n[2] = {0, 0};
  size_t pieces[2] = {0, 0};
  std::vector<uint64_t> part;          // digests of the pieces, array 0 then array 1
  std::atomic<size_t> next{0}, done{0};
  std::atomic<int> active{0};          // workers that hold a pointer to this job
  size_t total = 0;
};

struct HashPool {
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::thread> workers;
  std::vector<HashJob*> jobs;           // jobs in flight (guarded by mu)
  std::atomic<uint64_t> epoch{0};       // bumped with every new job
  bool quit = false;

  static void run_pieces(HashJob* j) {
    for (;;) {
      const size_t k = j->next.fetch_add(1, std::memory_order_relaxed);
      if (k >= j->total) return;
      const int a = k < j->pieces[0] ? 0 : 1;
      const size_t q = a ? k - j->pieces[0] : k;
      const size_t off = q * HASH_PIECE;
      const size_t len = std::min(HASH_PIECE, j->n[a] - off);
      j->part[k] = hash_piece(j->p[a] + off, len, (uint64_t)q);
      j->done.fetch_add(1, std::memory_order_release);
    }
  }
  void worker(unsigned index) {
    uint64_t seen = 0;
    std::vector<HashJob*> mine;
    for (;;) {
      // ONE worker spins briefly (the next job of a solve loop arrives within ~0.2 ms, and the caller's thread hashes
      // too); the others sleep on the condition variable at once: three to six threads spinning beside every solve
      // took CPU from the HIP runtime's own threads
      const auto spin = std::chrono::microseconds(index == 0 ? 300 : 0);
      const auto t0 = std::chrono::steady_clock::now();
      while (epoch.load(std::memory_order_acquire) == seen) {
        if (std::chrono::steady_clock::now() - t0 >= spin) {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return quit || epoch.load(std::memory_order_acquire) != seen; });
          if (quit) return;
          break;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (quit) return;
        seen = epoch.load(std::memory_order_acquire);
        mine = jobs;
        for (HashJob* j : mine) j->active.fetch_add(1, std::memory_order_relaxed);
      }
      for (HashJob* j : mine) {
        run_pieces(j);
        j->active.fetch_sub(1, std::memory_order_release);  // (the job may be freed from here on)
      }
    }
  }
  void ensure_started() {
    if (!workers.empty()) return;
    unsigned hw = std::thread::hardware_concurrency();
    unsigned nt = hw >= 32 ? 6 : (hw >= 8 ? 3 : 1);
    if (const char* env = std::getenv("SQD_HASH_THREADS")) nt = (unsigned)std::max(0, std::atoi(env));
    for (unsigned i = 0; i < nt; ++i) workers.emplace_back([this, i] { worker(i); });
    static std::once_flag once;
    std::call_once(once, [] { pthread_atfork(nullptr, nullptr, &HashPool::after_fork_in_child); });
  }
  // fork(): the child has the pool's memory but none of its threads, and `mu` may have been held by one of them at the
  // moment of the fork.  The child starts over with a fresh pool in the same storage (the old thread handles are
  // abandoned, not destroyed: their threads do not exist here).
  static void after_fork_in_child();
  void start(HashJob* j) {
    {
      std::lock_guard<std::mutex> lk(mu);
      ensure_started();
      jobs.push_back(j);
      epoch.fetch_add(1, std::memory_order_release);
    }
    cv.notify_all();
  }
  void finish(HashJob* j) {
    run_pieces(j);  // the waiting thread helps
    while (j->done.load(std::memory_order_acquire) < j->total) {
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    {
      std::lock_guard<std::mutex> lk(mu);
      for (size_t i = 0; i < jobs.size(); ++i)
        if (jobs[i] == j) {
          jobs.erase(jobs.begin() + (long)i);  // (no worker can pick it up any more)
          break;
        }
    }
    while (j->active.load(std::memory_order_acquire) != 0) {  // a late worker that found nothing left to do
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
  }
};
HashPool& hash_pool() {
  static HashPool* pool = new HashPool();  // (never destroyed: worker threads may outlive static destruction order)
  return *pool;
}
void HashPool::after_fork_in_child() { new (&hash_pool()) HashPool(); }
uint64_t fold_parts(const HashJob& j, int a) {
  const size_t b = a ? j.pieces[0] : 0;
  uint64_t h = HP3 ^ (uint64_t)j.n[a];
  for (size_t k = 0; k < j.pieces[a]; ++k) h = rotl64(h ^ j.part[b + k], 29) * HP1 + HP2;
  return avalanche(h);
}

}  // namespace
}  // namespace sqd

// digests of one or two byte ranges (p1 may be NULL); sqd_hash_start returns at once with a job handle, sqd_hash_finish
// waits (and helps) and releases the handle.  The ranges must stay unchanged and alive in between.  Jobs of several host
// threads may be in flight at once.
extern "C" __attribute__((visibility("default"))) int sqd_hash_start(const void* p0, size_t n0, const void* p1, size_t n1,
                                                                     void** job) {
  if ((!p0 && n0) || (!p1 && n1) || !job) {
    set_error("sqd_hash_start: bad argument");
    return SQD_ERR_INVALID;
  }
  HashJob* j = new HashJob();
  j->p[0] = static_cast<const unsigned char*>(p0);
  j->p[1] = static_cast<const unsigned char*>(p1);
  j->n[0] = n0;
  j->n[1] = n1;
  j->pieces[0] = (n0 + HASH_PIECE - 1) / HASH_PIECE;
  j->pieces[1] = (n1 + HASH_PIECE - 1) / HASH_PIECE;
  j->total = j->pieces[0] + j->pieces[1];
  j->part.assign(j->total, 0);
  if (j->total > 1) hash_pool().start(j);  // (a single piece: the finishing thread hashes it itself)
  *job = j;
  return SQD_OK;
}
extern "C" __attribute__((visibility("default"))) int sqd_hash_finish(void* job, unsigned long long* d0,
                                                                      unsigned long long* d1) {
  HashJob* j = static_cast<HashJob*>(job);
  if (!j) {
    set_error("sqd_hash_finish: no job");
    return SQD_ERR_INVALID;
  }
  if (j->total > 1) hash_pool().finish(j);
  else HashPool::run_pieces(j);
  if (d0) *d0 = fold_parts(*j, 0);
  if (d1) *d1 = fold_parts(*j, 1);
  delete j;
  return SQD_OK;
}

// Are both string lists what np.sort(np.unique(.)) would return, non-negative as int64, and of one Hamming weight each?
// (*ok = 1: the Python layer's _check_ci_strs -- reference fermion.py:1075-1097 -- has nothing to do or to raise; 0: it
// takes its numpy path, which raises the reference's errors or normalises the lists.)  Host code, 10^2 .. 10^4 words.
extern "C" __attribute__((visibility("default"))) int sqd_check_strings(const uint64_t* a, int64_t na, const uint64_t* b,
                                                                        int64_t nb, int* ok) {
  if (!ok || (!a && na) || (!b && nb) || na < 0 || nb < 0) {
    set_error("sqd_check_strings: bad argument");
    return SQD_ERR_INVALID;
  }
  auto good = [](const uint64_t* s, int64_t n) {
    if (n < 1 || (s[n - 1] >> 63)) return false;
    const int w = __builtin_popcountll(s[0]);
    bool fine = true;
    for (int64_t i = 1; i < n; ++i) fine &= (s[i] > s[i - 1]) & (__builtin_popcountll(s[i]) == w);
    return fine;
  };
  *ok = (good(a, na) && good(b, nb)) ? 1 : 0;
  return SQD_OK;
}
