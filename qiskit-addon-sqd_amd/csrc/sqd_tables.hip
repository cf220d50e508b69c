// CI-string addressing and per-subspace tables, built on the device.
//
// Replaces (see include/sqd_hip.h): pyscf selected_ci._all_linkstr_index
// (SCIcre_des_linkstr / SCIdes_des_linkstr), SelectedCI.make_hdiag and the per-call
// integral re-packing of selected_ci.contract_2e; reached from the reference at
// qiskit_addon_sqd/fermion.py:721-723 and :810-818.
//
// gfx950 design: the coupling structure of a *selected* string set is sparse and
// irregular, so it is enumerated, not searched: one wavefront per target string sweeps the
// sorted string table 64 entries at a time, classifies every pair with XOR + popcount
// (2 differing bits = single excitation, 4 = same-spin double), and compacts the hits with
// wave ballot + prefix popcount.  The links of a string therefore come out sorted by source
// address, which is this build's canonical (bit-exact, testable) order.  Integral values are
// attached by a second, fully occupied thread-per-link pass.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "sqd_common.h"

namespace sqd {

// ------------------------------------------------------------------ small helpers
int DevBuf::reserve(size_t bytes) {
  if (view) {
    p = nullptr;
    view = false;
  }
  if (bytes <= cap && p) return SQD_OK;
  if (p) {
    hipError_t e = hipFree(p);
    (void)e;
    p = nullptr;
    cap = 0;
  }
  size_t want = bytes + bytes / 4 + 256;
  hipError_t e = hipMalloc(&p, want);
  if (e != hipSuccess) {
    p = nullptr;
    set_error(std::string("hipMalloc(") + std::to_string(want) + ") failed: " + hipGetErrorString(e));
    return SQD_ERR_HIP;
  }
  cap = want;
  return SQD_OK;
}
void DevBuf::release() {
  if (view) {
    p = nullptr;
    view = false;
    return;
  }
  if (p) {
    hipError_t e = hipFree(p);
    (void)e;
  }
  p = nullptr;
  cap = 0;
}
void SpinTables::release() {
  DevBuf* all[] = {&strs, &e_str, &s_ptr, &d_ptr, &s_row, &d_row, &s_rec, &s_val, &d_src, &d_orb,
                   &d_val, &hs_ptr, &hs_src, &hs_val, &jrow, &jT, &es_sl, &ed_sl, &es_rec, &es_val, &ed_src, &ed_val,
                   &vs_cnt, &vs_own, &vs_start, &vd_cnt, &vd_own, &vd_start, &vs_chunk, &vd_chunk};
  for (DevBuf* b : all) b->release();
}

// ---- pinned staging arena (see sqd_ctx::stage_*)
static int stage_reset(sqd_ctx* c) {
  if (c->stage_pending) {  // copies of the previous set_subspace still read the arena
    SQD_HIP_CHECK(hipEventSynchronize(c->ev[1]));
    c->stage_pending = false;
  }
  if (c->stage_blocks.size() > 1) {  // grew last time: one block of the total size from now on
    for (void* p : c->stage_blocks) SQD_HIP_CHECK(hipHostFree(p));
    c->stage_blocks.clear();
    c->stage_cur = nullptr;
    c->stage_cap = 0;
    void* p = nullptr;
    SQD_HIP_CHECK(hipHostMalloc(&p, c->stage_total + (1 << 16), hipHostMallocDefault));
    c->stage_blocks.push_back(p);
    c->stage_cur = static_cast<char*>(p);
    c->stage_cap = c->stage_total + (1 << 16);
  }
  c->stage_off = 0;
  c->stage_total = 0;
  return SQD_OK;
}
static int stage_alloc(sqd_ctx* c, size_t bytes, void** out) {
  bytes = (bytes + 63) & ~size_t(63);
  c->stage_total += bytes;
  if (c->stage_off + bytes > c->stage_cap) {
    const size_t want = bytes * 2 > (size_t(1) << 20) ? bytes * 2 : (size_t(1) << 20);
    void* p = nullptr;
    SQD_HIP_CHECK(hipHostMalloc(&p, want, hipHostMallocDefault));
    c->stage_blocks.push_back(p);
    c->stage_cur = static_cast<char*>(p);
    c->stage_cap = want;
    c->stage_off = 0;
  }
  *out = c->stage_cur + c->stage_off;
  c->stage_off += bytes;
  return SQD_OK;
}
// host data -> pinned arena -> device, asynchronously on the context stream
static int stage_upload(sqd_ctx* c, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return SQD_OK;
  void* h = nullptr;
  SQD_TRY(stage_alloc(c, bytes, &h));
  std::memcpy(h, src, bytes);
  SQD_HIP_CHECK(hipMemcpyAsync(dst, h, bytes, hipMemcpyHostToDevice, c->stream));
  return SQD_OK;
}

__device__ inline int ctz64(uint64_t x) { return __ffsll((long long)x) - 1; }
__device__ inline uint64_t below_mask(int p) { return (p >= 64) ? ~0ull : ((1ull << p) - 1ull); }

// ------------------------------------------------------------------ integral tables
// eri_pp[tril(p,q)][tril(r,s)] = (pq|rs);  jm[i][j] = (ii|jj);  km[i][j] = (ij|ji)
__global__ void k_pack_eri(const double* __restrict__ eri4, int norb, int nnorb, double* __restrict__ eri_pp,
                           double* __restrict__ jm, double* __restrict__ km) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n4 = (int64_t)norb * norb * norb * norb;
  if (idx >= n4) return;
  const int s = idx % norb;
  const int r = (idx / norb) % norb;
  const int q = (idx / ((int64_t)norb * norb)) % norb;
  const int p = idx / ((int64_t)norb * norb * norb);
  const double v = eri4[idx];
  if (p >= q && r >= s) eri_pp[(int64_t)tril(p, q) * nnorb + tril(r, s)] = v;
  if (p == q && r == s) jm[p * norb + r] = v;
  if (p == s && q == r) km[p * norb + q] = v;
}

// ------------------------------------------------------------------ pair enumeration
// One wavefront per target string I.  pc = popcount(I ^ J): 2 -> single, 4 -> double.
__device__ inline void count_links_body(const uint64_t* __restrict__ strs, int64_t n, int64_t* __restrict__ cnt_s,
                                        int64_t* __restrict__ cnt_d) {
  const int lane = threadIdx.x & 63;
  const int64_t I = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (I >= n) return;  // whole wave leaves together
  const uint64_t sI = strs[I];
  int64_t cs = 0, cd = 0;
  for (int64_t j0 = 0; j0 < n; j0 += 64) {
    const int64_t J = j0 + lane;
    int pc = 0;
    if (J < n) pc = __popcll(sI ^ strs[J]);
    cs += __popcll(__ballot(pc == 2));
    cd += __popcll(__ballot(pc == 4));
  }
  if (lane == 0) {
    cnt_s[I] = cs;
    cnt_d[I] = cd;
  }
}

__device__ inline void fill_links_body(const uint64_t* __restrict__ strs, int64_t n, const int64_t* __restrict__ s_ptr,
                                       const int64_t* __restrict__ d_ptr, SRec* __restrict__ s_rec,
                                       uint32_t* __restrict__ s_row, uint32_t* __restrict__ d_src,
                                       uint32_t* __restrict__ d_row) {
  const int lane = threadIdx.x & 63;
  const int64_t I = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (I >= n) return;
  const uint64_t sI = strs[I];
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int64_t ps = s_ptr[I], pd = d_ptr[I];
  for (int64_t j0 = 0; j0 < n; j0 += 64) {
    const int64_t J = j0 + lane;
    int pc = 0;
    if (J < n) pc = __popcll(sI ^ strs[J]);
    const unsigned long long ms = __ballot(pc == 2);
    const unsigned long long md = __ballot(pc == 4);
    if (pc == 2) {
      const int64_t pos = ps + __popcll(ms & lt);
      s_rec[pos].src = (uint32_t)J;
      s_row[pos] = (uint32_t)I;
    }
    if (pc == 4) {
      const int64_t pos = pd + __popcll(md & lt);
      d_src[pos] = (uint32_t)J;
      d_row[pos] = (uint32_t)I;
    }
    ps += __popcll(ms);
    pd += __popcll(md);
  }
}

// ------------------------------------------------------------------ link decoration
// |I> = sign a+_cre a_des |J>;  value = sign * (h[cre,des] + sum_{k in J, k != des} (cre des|kk) - (cre k|k des))
__device__ inline void decorate_singles_body(const uint64_t* __restrict__ strs, int64_t n_s,
                                             const uint32_t* __restrict__ s_row, SRec* __restrict__ s_rec,
                                             double* __restrict__ s_val, const double* __restrict__ h1,
                                             const double* __restrict__ eri4, int norb) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_s) return;
  const uint64_t I = strs[s_row[l]];
  const uint32_t src = s_rec[l].src;
  const uint64_t J = strs[src];
  const uint64_t x = I ^ J;
  const int a = ctz64(x & I);  // created
  const int b = ctz64(x & J);  // annihilated
  const int lo = a < b ? a : b, hi = a < b ? b : a;
  const uint64_t between = below_mask(hi) & ~below_mask(lo + 1);
  const int neg = __popcll(J & between) & 1;
  const int64_t n1 = norb, n2 = n1 * norb, n3 = n2 * norb;
  double v = h1[a * norb + b];
  uint64_t occ = J & ~(1ull << b);
  while (occ) {
    const int k = ctz64(occ);
    occ &= occ - 1;
    v += eri4[a * n3 + b * n2 + k * n1 + k] - eri4[a * n3 + k * n2 + k * n1 + b];
  }
  s_val[l] = neg ? -v : v;
  const uint32_t widx = 2u * tril(a, b) + (a > b ? 1u : 0u);
  s_rec[l].meta = widx | ((uint32_t)a << 13) | ((uint32_t)b << 19) | ((uint32_t)neg << 31);
}

// |I> = sign a+_p a+_r a_s a_q |J>, p>r, q>s;  value = sign * ((pq|rs) - (ps|rq))
__device__ inline void decorate_doubles_body(const uint64_t* __restrict__ strs, int64_t n_d,
                                             const uint32_t* __restrict__ d_row, const uint32_t* __restrict__ d_src,
                                             uint32_t* __restrict__ d_orb, double* __restrict__ d_val,
                                             const double* __restrict__ eri4, int norb) {
  const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (l >= n_d) return;
  const uint64_t I = strs[d_row[l]];
  const uint64_t J = strs[d_src[l]];
  const uint64_t x = I ^ J;
  uint64_t cre = x & I, des = x & J;
  const int r = ctz64(cre);
  cre &= cre - 1;
  const int p = ctz64(cre);
  const int s = ctz64(des);
  des &= des - 1;
  const int q = ctz64(des);
  // apply a_q, a_s, a+_r, a+_p in that order, collecting parities
  uint64_t st = J;
  int par = __popcll(st & below_mask(q));
  st ^= 1ull << q;
  par += __popcll(st & below_mask(s));
  st ^= 1ull << s;
  par += __popcll(st & below_mask(r));
  st |= 1ull << r;
  par += __popcll(st & below_mask(p));
  const int neg = par & 1;
  const int64_t n1 = norb, n2 = n1 * norb, n3 = n2 * norb;
  const double v = eri4[p * n3 + q * n2 + r * n1 + s] - eri4[p * n3 + s * n2 + r * n1 + q];
  d_val[l] = neg ? -v : v;
  d_orb[l] = (uint32_t)p | ((uint32_t)r << 6) | ((uint32_t)q << 12) | ((uint32_t)s << 18) | ((uint32_t)neg << 31);
}

// ------------------------------------------------------------------ per-string tables
// e_str[I] = sum_{i in I} h_ii + 1/2 sum_{i,j in I} (J_ij - K_ij)
__device__ inline void string_energy_body(const uint64_t* __restrict__ strs, int64_t n, const double* __restrict__ h1,
                                          const double* __restrict__ jm, const double* __restrict__ km, int norb,
                                          double* __restrict__ e_str) {
  // one wavefront per string: lane l takes the orbital pairs (i, j) = (l / nocc, l % nocc), l += 64;
  // the shuffle tree adds them in fixed order
  const int lane = threadIdx.x & 63;
  const int64_t I = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (I >= n) return;
  const uint64_t s = strs[I];
  const int nocc = __popcll(s);
  double e = 0.0;
  for (int p = lane; p < nocc * nocc; p += 64) {
    const int a = p / nocc, b = p % nocc;
    uint64_t t = s;
    for (int k = 0; k < a; ++k) t &= t - 1;
    const int i = ctz64(t);
    t = s;
    for (int k = 0; k < b; ++k) t &= t - 1;
    const int j = ctz64(t);
    e += 0.5 * (jm[i * norb + j] - km[i * norb + j]);
    if (a == b) e += h1[i * norb + i];
  }
  for (int off = 32; off > 0; off >>= 1) e += __shfl_down(e, off);
  if (lane == 0) e_str[I] = e;
}

// J[I][pair] = sum_{k in I} (pair|kk).  transposed == 0: out[I*nnorb + pair]; else out[pair*n + I]
__device__ inline void jtable_body(const uint64_t* __restrict__ strs, int64_t n, const double* __restrict__ eri_pp,
                                   int nnorb, int transposed, double* __restrict__ out) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * nnorb) return;
  int64_t I, pair;
  if (transposed) {
    pair = idx / n;
    I = idx % n;
  } else {
    I = idx / nnorb;
    pair = idx % nnorb;
  }
  uint64_t occ = strs[I];
  double v = 0.0;
  while (occ) {
    const int k = ctz64(occ);
    occ &= occ - 1;
    v += eri_pp[pair * nnorb + (int64_t)k * (k + 1) / 2 + k];
  }
  out[idx] = v;
}

// ---- both spins in one launch (gridDim.y = 2): at the sizes of one subsample batch these kernels run a few
// microseconds each, less than it costs the host to enqueue one, so halving the launches is what counts
struct SpinLinkArgs {
  const uint64_t* strs;
  int64_t n, n_s, n_d;
  int64_t *cnt_s, *cnt_d;        // pass 1
  int64_t *s_ptr, *d_ptr;        // pass 2
  SRec* s_rec;
  uint32_t *s_row, *d_src, *d_row, *d_orb;
  double *s_val, *d_val;
  double* e_str;
  double* jtab;                  // jrow (alpha: [I][pair]) or jT (beta: [pair][I])
  int transposed;
};
struct SpinLinkArgs2 {
  SpinLinkArgs a[2];
};
__global__ void k_count_links2(const SpinLinkArgs2 p) {
  const SpinLinkArgs& a = p.a[blockIdx.y];
  count_links_body(a.strs, a.n, a.cnt_s, a.cnt_d);
}
__global__ void k_fill_links2(const SpinLinkArgs2 p) {
  const SpinLinkArgs& a = p.a[blockIdx.y];
  fill_links_body(a.strs, a.n, a.s_ptr, a.d_ptr, a.s_rec, a.s_row, a.d_src, a.d_row);
}
// singles and doubles of both spins: blockIdx.y = 2 * spin + (0 singles | 1 doubles)
__global__ void k_decorate_links2(const SpinLinkArgs2 p, const double* __restrict__ h1, const double* __restrict__ eri4,
                                  int norb) {
  const SpinLinkArgs& a = p.a[blockIdx.y >> 1];
  if ((blockIdx.y & 1) == 0)
    decorate_singles_body(a.strs, a.n_s, a.s_row, a.s_rec, a.s_val, h1, eri4, norb);
  else
    decorate_doubles_body(a.strs, a.n_d, a.d_row, a.d_src, a.d_orb, a.d_val, eri4, norb);
}
// per-string energies (blockIdx.y = spin) and occupation tables (blockIdx.y = 2 + spin)
__global__ void k_string_tables2(const SpinLinkArgs2 p, const double* __restrict__ h1, const double* __restrict__ jm,
                                 const double* __restrict__ km, const double* __restrict__ eri_pp, int norb, int nnorb) {
  const SpinLinkArgs& a = p.a[blockIdx.y & 1];
  if (blockIdx.y < 2)
    string_energy_body(a.strs, a.n, h1, jm, km, norb, a.e_str);
  else
    jtable_body(a.strs, a.n, eri_pp, nnorb, a.transposed, a.jtab);
}

// hdiag[A,B] = e_a[A] + e_b[B] + sum_{i in A} JT_b[tril(i,i)][B]
__global__ void k_hdiag(const uint64_t* __restrict__ strs_a, const double* __restrict__ e_a,
                        const double* __restrict__ e_b, const double* __restrict__ jT_b, int64_t na, int64_t nb,
                        double* __restrict__ hdiag) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= na * nb) return;
  const int64_t A = idx / nb, B = idx - A * nb;
  uint64_t occ = strs_a[A];
  double v = e_a[A] + e_b[B];
  while (occ) {
    const int i = ctz64(occ);
    occ &= occ - 1;
    v += jT_b[((int64_t)i * (i + 1) / 2 + i) * nb + B];
  }
  hdiag[A * nb + B] = v;
}

// ------------------------------------------------------------------ capped sliced ELL (column role)
// Lanes of the sigma kernel would map to beta strings, so one string with hundreds of links (the
// Hartree-Fock neighbourhood) would stall its whole wavefront and leave the LDS pipe running mostly
// masked-off lanes.  Every list is therefore cut into *virtual rows* of at most CAP links that are
// processed by whichever thread comes next, in an order that keeps wavefronts uniform: all full
// rows first (grouped by owner), then the tails by descending length.  Row partial sums meet in LDS
// and each string adds its own rows (one contiguous run + one tail) in fixed order.  Storage is
// sliced ELL over the ordered rows: entry (k, lane) of slice b at sl[b] + 64 k + lane.  Descriptors
// are computed on the host from the CSR pointers (they arrive with the one synchronisation of
// set_subspace); the fill runs on the device.
__global__ void k_fill_vell_singles(int64_t nv, const int32_t* __restrict__ vcnt, const int64_t* __restrict__ vstart,
                                    const int64_t* __restrict__ sl, const SRec* __restrict__ rec,
                                    const double* __restrict__ val, SRec* __restrict__ erec,
                                    double* __restrict__ eval) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  const int64_t base = sl[v >> 6] + (v & 63);
  const int64_t p0 = vstart[v];
  const int cnt = vcnt[v];
  for (int k = 0; k < cnt; ++k) {
    erec[base + (int64_t)k * 64] = rec[p0 + k];
    eval[base + (int64_t)k * 64] = val[p0 + k];
  }
}
__global__ void k_fill_vell_doubles(int64_t nv, const int32_t* __restrict__ vcnt, const int64_t* __restrict__ vstart,
                                    const int64_t* __restrict__ sl, const uint32_t* __restrict__ src,
                                    const double* __restrict__ val, uint32_t* __restrict__ esrc,
                                    double* __restrict__ eval) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  const int64_t base = sl[v >> 6] + (v & 63);
  const int64_t p0 = vstart[v];
  const int cnt = vcnt[v];
  for (int k = 0; k < cnt; ++k) {
    esrc[base + (int64_t)k * 64] = src[p0 + k];
    eval[base + (int64_t)k * 64] = val[p0 + k];
  }
}

// host side of the above: virtual-row descriptors from a CSR pointer array
// (owners are taken one column chunk at a time; a chunk's rows are contiguous and padded with empty
// rows to a whole 64-row slice, so that slices never straddle chunks)
static void make_vrows(const int64_t* ptr, int64_t n, int cap, int64_t chunk_cols, VRowsHost& out) {
  out.own.assign(3 * n, 0);
  out.vcnt.clear();
  out.vstart.clear();
  out.chunk.clear();
  out.nv_max = 0;
  std::vector<std::pair<int32_t, int64_t>> tails;  // (length, owner)
  for (int64_t c0 = 0; c0 < n; c0 += chunk_cols) {
    const int64_t c1 = (c0 + chunk_cols < n) ? c0 + chunk_cols : n;
    const int64_t v0 = (int64_t)out.vcnt.size();
    out.chunk.push_back((int32_t)v0);
    // full rows, grouped by owner
    for (int64_t i = c0; i < c1; ++i) {
      const int64_t cnt = ptr[i + 1] - ptr[i];
      const int64_t nfull = cnt / cap;
      out.own[3 * i + 0] = (int32_t)out.vcnt.size();
      out.own[3 * i + 1] = (int32_t)nfull;
      out.own[3 * i + 2] = -1;
      for (int64_t j = 0; j < nfull; ++j) {
        out.vcnt.push_back(cap);
        out.vstart.push_back(ptr[i] + j * cap);
      }
    }
    // tails by descending length (stable in owner)
    tails.clear();
    for (int64_t i = c0; i < c1; ++i) {
      const int64_t rem = (ptr[i + 1] - ptr[i]) % cap;
      if (rem > 0) tails.emplace_back((int32_t)rem, i);
    }
    std::stable_sort(tails.begin(), tails.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    for (const auto& t : tails) {
      const int64_t i = t.second;
      out.own[3 * i + 2] = (int32_t)out.vcnt.size();
      out.vcnt.push_back(t.first);
      out.vstart.push_back(ptr[i + 1] - t.first);
    }
    if (c1 < n)
      while (out.vcnt.size() % 64) {  // empty padding rows
        out.vcnt.push_back(0);
        out.vstart.push_back(0);
      }
    const int64_t used = (int64_t)out.vcnt.size() - v0;
    if (used > out.nv_max) out.nv_max = used;
  }
  out.chunk.push_back((int32_t)out.vcnt.size());
  out.nv = (int64_t)out.vcnt.size();
  const int64_t nsl = (out.nv + 63) / 64;
  out.sl.assign(nsl + 1, 0);
  for (int64_t b = 0; b < nsl; ++b) {
    int32_t w = 0;
    for (int64_t v = b * 64; v < b * 64 + 64 && v < out.nv; ++v) w = out.vcnt[v] > w ? out.vcnt[v] : w;
    out.sl[b + 1] = out.sl[b] + 64 * (int64_t)w;
  }
  out.total = out.sl[nsl];
}

// merged same-spin CSR: row i = its single links (value incl. sign) followed by its double links
__global__ void k_merge_hs(int64_t n, const int64_t* __restrict__ s_ptr, const int64_t* __restrict__ d_ptr,
                           const SRec* __restrict__ s_rec, const double* __restrict__ s_val,
                           const uint32_t* __restrict__ d_src, const double* __restrict__ d_val,
                           int64_t* __restrict__ hs_ptr, uint32_t* __restrict__ hs_src, double* __restrict__ hs_val) {
  // one wavefront per row (rows of the Hartree-Fock neighbourhood hold hundreds of links)
  const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (i > n) return;
  const int64_t o = s_ptr[i] + d_ptr[i];
  if (lane == 0) hs_ptr[i] = o;
  if (i == n) return;
  const int64_t s0 = s_ptr[i], ns = s_ptr[i + 1] - s0, d0 = d_ptr[i], nd = d_ptr[i + 1] - d0;
  for (int64_t k = lane; k < ns; k += 64) {
    hs_src[o + k] = s_rec[s0 + k].src;
    hs_val[o + k] = s_val[s0 + k];
  }
  for (int64_t k = lane; k < nd; k += 64) {
    hs_src[o + ns + k] = d_src[d0 + k];
    hs_val[o + ns + k] = d_val[d0 + k];
  }
}

// ------------------------------------------------------------------ sigma work list (host)
// The sigma kernel runs one workgroup per work item so that a few highly connected strings (the
// Hartree-Fock neighbourhood) do not serialise the launch:
//   type 0  own row: diagonal, beta links on the LDS-staged row, first L0 same-spin alpha links
//   type 1  a batch of <= K alpha single links (K source rows + K integral rows staged in LDS)
//   type 2  a chunk of <= L further same-spin alpha links (unit-stride row AXPYs)
// A row with a single item writes sigma directly; otherwise items write partial rows that
// k_sigma_reduce adds in fixed order.
// Launch geometry and LDS plan of k_sigma for a given virtual-row layout.
//   T threads cover the row in R strides; the beta virtual rows are spread over all T threads.
//   LDS of one workgroup: [singles partials | penw | region], where the region holds K staged (C row +
//   integral row) pairs for an alpha-single batch, or -- for an own-row item, which stages one pair -- that
//   pair followed by the doubles' partial sums.  The two uses overlap: the allocation is the LARGER of them.
//   wgs = workgroups resident per CU (LDS and the 32-wave limit): what the layout search maximises.
struct SigmaPlan {
  int T = 64, R = 1, K = 1, nb_pad = 0, wgs = 0;
  size_t shmem = 0;
};
static SigmaPlan plan_sigma(const sqd_ctx* c, int64_t nb, const VRowsHost& vs, const VRowsHost& vd, int kmax) {
  SigmaPlan p;
  // threads per workgroup (measured on MI355X, profiles/r01/sigma_geometry_sweep.txt): 512 up to
  // nb = 2048, 1024 beyond; never more than the row or the virtual-row lists can occupy
  const int64_t nvmax = vs.nv > vd.nv ? vs.nv : vd.nv;
  const int64_t want = nb > nvmax ? nb : nvmax;
  int T = (int)(((want + 63) / 64) * 64);
  const int tmax = (nb <= 2048) ? 512 : 1024;
  if (T > tmax) T = tmax;
  if (T < 64) T = 64;
  if (const char* env = std::getenv("SQD_SIGMA_T")) {  // tuning hook
    const int v = (std::atoi(env) / 64) * 64;
    if (v >= 64 && v <= 1024) T = v;
  }
  p.nb_pad = (int)((nb + 1) & ~int64_t(1));
  const size_t w2_bytes = (size_t)((c->nnorb + 1) & ~1) * 8;
  const size_t row_bytes = (size_t)p.nb_pad * 8 + w2_bytes;  // one C row + one integral row
  const size_t ps_bytes = (size_t)((c->sig_ps + 1) & ~int64_t(1)) * 8 + 16;  // singles partials + penw
  const size_t pd_bytes = (size_t)c->sig_pd * 8;
  const size_t budget = (size_t)c->lds_bytes - 8 * 1024;
  auto plan_bytes = [&](int k) {
    const size_t batch = (size_t)k * row_bytes, own = row_bytes + pd_bytes;
    return ps_bytes + (batch > own ? batch : own) + 64;
  };
  if (c->sig_lds_rows) {
    p.R = (int)((nb + T - 1) / T);
    int K = kmax;  // more than 4 links per batch lengthen the batch without saving launches (geometry sweep)
    while (K > 1 && ((size_t)K * row_bytes > 96 * 1024 || plan_bytes(K) > budget)) --K;
    if (const char* env = std::getenv("SQD_SIGMA_K")) {  // tuning hook
      const int v = std::atoi(env);
      if (v >= 1 && v <= K) K = v;
    }
    p.K = K;
    p.shmem = plan_bytes(K);
  } else {
    // rows stay in global memory: one link per batch, workgroups of one column chunk
    T = (int)(c->sig_chunk < 1024 ? c->sig_chunk : 1024);
    p.R = (int)((c->sig_chunk + T - 1) / T);
    p.K = 1;
    p.shmem = ps_bytes + w2_bytes + pd_bytes + 64;
  }
  p.T = T;
  const int by_lds = (int)((size_t)c->lds_bytes / p.shmem), by_waves = 32 / (T / 64);
  p.wgs = by_lds < by_waves ? by_lds : by_waves;
  return p;
}

static int build_sigma_work(sqd_ctx* c) {
  const int64_t na = c->na, nb = c->nb;
  const SigmaPlan plan = plan_sigma(c, nb, c->hv_s, c->hv_d, c->sig_kmax);
  const int T = plan.T, R = plan.R, K = plan.K;
  if (plan.shmem > (size_t)c->lds_bytes || R > 16) {
    set_error("beta string count " + std::to_string(nb) + " exceeds the sigma kernel's row geometry");
    return SQD_ERR_LIMIT;
  }
  c->sig_T = T;
  c->sig_R = R;
  c->sig_K = K;
  c->sig_nb_pad = plan.nb_pad;
  c->sig_shmem = plan.shmem;
  // same-spin links folded into the own-row item: sparse sets (few links per row) take all of them there
  // and need no partial rows / reduce launch; well-connected sets keep the own-row item short
  const int64_t hs_total = c->h_sptr[na] + c->h_dptr[na];
  int L0 = (hs_total <= 24 * na) ? 32 : 16, L = 32;
  if (const char* env = std::getenv("SQD_SIGMA_L")) {  // test hook: tiny chunks => many AXPY items per row
    const int v = std::atoi(env);
    if (v >= 1 && v <= 32) L0 = L = v;
  }
  if (const char* env = std::getenv("SQD_SIGMA_LCHUNK")) {  // tuning hook: links per AXPY item
    const int v = std::atoi(env);
    if (v >= 1 && v <= 4096) L = v;
  }
  if (const char* env = std::getenv("SQD_SIGMA_L0")) {  // tuning hook: links folded into the own-row item
    const int v = std::atoi(env);
    if (v >= 0 && v <= 4096) L0 = v;
  }
  std::vector<WorkItem>& items = c->h_items;
  std::vector<MultiRow>& multi = c->h_multi;
  items.clear();
  multi.clear();
  int32_t nslots = 0;
  std::vector<WorkItem> row;
  for (int64_t A = 0; A < na; ++A) {
    row.clear();
    const int64_t s0 = c->h_sptr[A], s1 = c->h_sptr[A + 1];
    const int64_t h0 = s0 + c->h_dptr[A], h1 = s1 + c->h_dptr[A + 1];
    WorkItem own{h0, (uint32_t)A, 0, (uint16_t)((h1 - h0 < L0) ? (h1 - h0) : L0), -1, 0};
    row.push_back(own);
    for (int64_t l = s0; l < s1; l += K)
      row.push_back(WorkItem{l, (uint32_t)A, 1, (uint16_t)((s1 - l < K) ? (s1 - l) : K), -1, 0});
    for (int64_t l = h0 + L0; l < h1; l += L)
      row.push_back(WorkItem{l, (uint32_t)A, 2, (uint16_t)((h1 - l < L) ? (h1 - l) : L), -1, 0});
    if (row.size() > 1) {  // several items: partial rows + fixed-order reduce
      multi.push_back(MultiRow{(uint32_t)A, nslots, (int32_t)row.size()});
      for (auto& it : row) it.slot = nslots++;
    }
    items.insert(items.end(), row.begin(), row.end());
  }
  // heaviest item class first (LDS-staged batches), then own rows, then AXPY chunks
  std::stable_sort(items.begin(), items.end(), [](const WorkItem& a, const WorkItem& b) {
    auto rank = [](const WorkItem& w) { return w.type == 1 ? 0 : (w.type == 0 ? 1 : 2); };
    return rank(a) < rank(b);
  });
  c->n_items = (int64_t)items.size();
  c->n_multi = (int64_t)multi.size();
  c->n_slots = nslots;
  if (std::getenv("SQD_DEBUG_GEOM"))
    std::fprintf(stderr,
                 "[sqd geom] na %lld nb %lld T %d R %d K %d lds_rows %d chunks %d cap %d nvs %lld/%lld nvd %lld/%lld shmem %zu items %zu "
                 "multi %zu slots %d\n",
                 (long long)na, (long long)nb, T, R, K, (int)c->sig_lds_rows, c->sig_nchunks, c->sp[1].cap,
                 (long long)c->hv_s.nv_max, (long long)c->sig_ps, (long long)c->hv_d.nv_max, (long long)c->sig_pd, c->sig_shmem, items.size(), multi.size(), nslots);
  // (the lists travel to the device inside the descriptor blob of build_subspace: one copy, not three)
  SQD_TRY(c->sig_partial.reserve((size_t)nslots * nb * 8 + 8));
  return SQD_OK;
}

// ------------------------------------------------------------------ host orchestration
static inline unsigned nblk(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

int build_integral_tables(sqd_ctx* c, const double* h1, const double* eri) {
  const int norb = c->norb;
  const int nnorb = norb * (norb + 1) / 2;
  c->nnorb = nnorb;
  const int64_t n2 = (int64_t)norb * norb, n4 = n2 * n2;
  SQD_TRY(c->h1.reserve(n2 * 8));
  SQD_TRY(c->eri4.reserve(n4 * 8));
  SQD_TRY(c->eri_pp.reserve((int64_t)nnorb * nnorb * 8));
  SQD_TRY(c->jm.reserve(n2 * 8));
  SQD_TRY(c->km.reserve(n2 * 8));
  SQD_HIP_CHECK(hipMemcpyAsync(c->h1.p, h1, n2 * 8, hipMemcpyHostToDevice, c->stream));
  SQD_HIP_CHECK(hipMemcpyAsync(c->eri4.p, eri, n4 * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_pack_eri, dim3(nblk(n4, 256)), dim3(256), 0, c->stream, c->eri4.as<double>(), norb, nnorb,
                     c->eri_pp.as<double>(), c->jm.as<double>(), c->km.as<double>());
  SQD_HIP_CHECK(hipGetLastError());
  SQD_HIP_CHECK(hipStreamSynchronize(c->stream));
  return SQD_OK;
}

static int validate_strings(const uint64_t* s, int64_t n, int norb, const char* which, int* nocc) {
  if (n <= 0 || s == nullptr) {
    set_error(std::string("empty ") + which + " string list");
    return SQD_ERR_INVALID;
  }
  if (n > 0xffffffffll) {
    set_error("more than 2^32 strings per spin is not supported");
    return SQD_ERR_LIMIT;
  }
  const int h0 = __builtin_popcountll(s[0]);
  for (int64_t i = 0; i < n; ++i) {
    if (norb < 64 && (s[i] >> norb)) {
      set_error(std::string(which) + " CI string in index " + std::to_string(i) + " has a bit at or above norb");
      return SQD_ERR_INVALID;
    }
    const int h = __builtin_popcountll(s[i]);
    if (h != h0) {
      set_error(std::string(which) + " CI string in index 0 has hamming weight " + std::to_string(h0) +
                ", but CI string in index " + std::to_string(i) + " has hamming weight " + std::to_string(h) + ".");
      return SQD_ERR_INVALID;
    }
    if (i > 0 && !(s[i - 1] < s[i])) {
      set_error(std::string(which) + " CI strings must be strictly ascending (index " + std::to_string(i) + ")");
      return SQD_ERR_INVALID;
    }
  }
  *nocc = h0;
  return SQD_OK;
}

// four independent exclusive scans in one launch (one workgroup each)
struct ScanJobs {
  const int64_t* in[4];
  int64_t* out[4];
  int64_t n[4];
};
__global__ void k_exclusive_scan4(const ScanJobs jobs) {
  __shared__ int64_t sums[1024];
  const int64_t* __restrict__ in = jobs.in[blockIdx.x];
  int64_t* __restrict__ out = jobs.out[blockIdx.x];
  const int64_t n = jobs.n[blockIdx.x];
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

int build_subspace(sqd_ctx* c, const uint64_t* sa, int64_t na, const uint64_t* sb, int64_t nb) {
  c->have_subspace = false;
  c->have_solution = false;
  int nocc[2];
  SQD_TRY(validate_strings(sa, na, c->norb, "Spin-up", &nocc[0]));
  SQD_TRY(validate_strings(sb, nb, c->norb, "Spin-down", &nocc[1]));
  const int norb = c->norb, nnorb = c->nnorb;
  hipStream_t st = c->stream;
  SQD_TRY(stage_reset(c));
  SQD_HIP_CHECK(hipEventRecord(c->ev[0], st));

  const uint64_t* hs[2] = {sa, sb};
  const int64_t ns[2] = {na, nb};
  int64_t maxn = na > nb ? na : nb;
  SQD_TRY(c->scratch.reserve((size_t)(4 * maxn + 4 * (maxn / 64 + 2) + 64) * 8));
  for (int s = 0; s < 2; ++s) {
    SpinTables& t = c->sp[s];
    t.n = ns[s];
    t.nocc = nocc[s];
    t.n_slices = (t.n + 63) / 64;
    SQD_TRY(t.strs.reserve(t.n * 8));
    SQD_TRY(stage_upload(c, t.strs.p, hs[s], (size_t)t.n * 8));
  }
  // pass 1: counts + CSR pointers for both spins, one host sync for the totals
  // all four CSR pointer arrays live in one device buffer so that ONE copy brings them (and with them
  // every total the host needs) back: the sigma work list (alpha) and the capped-ELL geometry (beta)
  // are cut on the host from these pointers
  const int64_t nptr = 2 * (na + 1) + 2 * (nb + 1);
  SQD_TRY(c->ptrs.reserve((size_t)nptr * 8));
  {
    int64_t* base = c->ptrs.as<int64_t>();
    c->sp[0].s_ptr.set_view(base);
    c->sp[0].d_ptr.set_view(base + (na + 1));
    c->sp[1].s_ptr.set_view(base + 2 * (na + 1));
    c->sp[1].d_ptr.set_view(base + 2 * (na + 1) + (nb + 1));
  }
  SpinLinkArgs2 la;  // both spins' arguments, filled in as the buffers come to exist
  std::memset(&la, 0, sizeof(la));
  {
    int64_t* d_cnt = c->scratch.as<int64_t>();  // [cnt_s_a | cnt_d_a | cnt_s_b | cnt_d_b], maxn each
    ScanJobs jobs;
    for (int s = 0; s < 2; ++s) {
      SpinTables& t = c->sp[s];
      int64_t* cnt_s = d_cnt + (2 * s) * maxn;
      int64_t* cnt_d = d_cnt + (2 * s + 1) * maxn;
      la.a[s].strs = t.strs.as<uint64_t>();
      la.a[s].n = t.n;
      la.a[s].cnt_s = cnt_s;
      la.a[s].cnt_d = cnt_d;
      la.a[s].s_ptr = t.s_ptr.as<int64_t>();
      la.a[s].d_ptr = t.d_ptr.as<int64_t>();
      if (s == 1) hipLaunchKernelGGL(k_count_links2, dim3(nblk(maxn, 4), 2), dim3(256), 0, st, la);
      jobs.in[2 * s] = cnt_s;
      jobs.out[2 * s] = t.s_ptr.as<int64_t>();
      jobs.n[2 * s] = t.n;
      jobs.in[2 * s + 1] = cnt_d;
      jobs.out[2 * s + 1] = t.d_ptr.as<int64_t>();
      jobs.n[2 * s + 1] = t.n;
    }
    hipLaunchKernelGGL(k_exclusive_scan4, dim3(4), dim3(256), 0, st, jobs);
    SQD_HIP_CHECK(hipGetLastError());
  }
  void* h_ptrs = nullptr;  // pinned: the copy is asynchronous, the host waits on the event below
  SQD_TRY(stage_alloc(c, (size_t)nptr * 8, &h_ptrs));
  SQD_HIP_CHECK(hipMemcpyAsync(h_ptrs, c->ptrs.p, (size_t)nptr * 8, hipMemcpyDeviceToHost, st));
  SQD_HIP_CHECK(hipEventRecord(c->ev_aux, st));
  // everything that needs only the strings is queued BEHIND the copy and runs while the host waits for the
  // pointers and cuts the work lists: per-string energies, occupation tables, the diagonal
  SQD_TRY(c->hdiag.reserve((size_t)na * nb * 8));
  for (int s = 0; s < 2; ++s) {
    SpinTables& t = c->sp[s];
    SQD_TRY(t.e_str.reserve(t.n * 8));
    DevBuf& jt = (s == 0) ? t.jrow : t.jT;  // alpha: J[I][pair] (row role); beta: transposed (column role)
    SQD_TRY(jt.reserve(t.n * nnorb * 8));
    la.a[s].e_str = t.e_str.as<double>();
    la.a[s].jtab = jt.as<double>();
    la.a[s].transposed = s;
  }
  {
    const unsigned gx_e = nblk(maxn, 4), gx_j = nblk(maxn * nnorb, 256);
    hipLaunchKernelGGL(k_string_tables2, dim3(gx_e > gx_j ? gx_e : gx_j, 4), dim3(256), 0, st, la, c->h1.as<double>(),
                       c->jm.as<double>(), c->km.as<double>(), c->eri_pp.as<double>(), norb, nnorb);
  }
  hipLaunchKernelGGL(k_hdiag, dim3(nblk(na * nb, 256)), dim3(256), 0, st, c->sp[0].strs.as<uint64_t>(),
                     c->sp[0].e_str.as<double>(), c->sp[1].e_str.as<double>(), c->sp[1].jT.as<double>(), na, nb,
                     c->hdiag.as<double>());
  SQD_HIP_CHECK(hipGetLastError());
  SQD_HIP_CHECK(hipEventSynchronize(c->ev_aux));
  c->h_sptr = static_cast<const int64_t*>(h_ptrs);
  c->h_dptr = c->h_sptr + (na + 1);
  c->h_sptr_b = c->h_dptr + (na + 1);
  c->h_dptr_b = c->h_sptr_b + (nb + 1);
  const int64_t tot[4] = {c->h_sptr[na], c->h_dptr[na], c->h_sptr_b[nb], c->h_dptr_b[nb]};
  // pass 2: fill + decorate
  for (int s = 0; s < 2; ++s) {
    SpinTables& t = c->sp[s];
    t.n_s = tot[2 * s];
    t.n_d = tot[2 * s + 1];
    SQD_TRY(t.s_rec.reserve((size_t)t.n_s * sizeof(SRec)));
    SQD_TRY(t.s_row.reserve((size_t)t.n_s * 4));
    SQD_TRY(t.s_val.reserve((size_t)t.n_s * 8));
    SQD_TRY(t.d_src.reserve((size_t)t.n_d * 4));
    SQD_TRY(t.d_row.reserve((size_t)t.n_d * 4));
    SQD_TRY(t.d_orb.reserve((size_t)t.n_d * 4));
    SQD_TRY(t.d_val.reserve((size_t)t.n_d * 8));
    la.a[s].n_s = t.n_s;
    la.a[s].n_d = t.n_d;
    la.a[s].s_rec = t.s_rec.as<SRec>();
    la.a[s].s_row = t.s_row.as<uint32_t>();
    la.a[s].s_val = t.s_val.as<double>();
    la.a[s].d_src = t.d_src.as<uint32_t>();
    la.a[s].d_row = t.d_row.as<uint32_t>();
    la.a[s].d_orb = t.d_orb.as<uint32_t>();
    la.a[s].d_val = t.d_val.as<double>();
  }
  {
    int64_t maxl = 0;
    for (int64_t v : tot) maxl = v > maxl ? v : maxl;
    if (maxl > 0) {
      hipLaunchKernelGGL(k_fill_links2, dim3(nblk(maxn, 4), 2), dim3(256), 0, st, la);
      hipLaunchKernelGGL(k_decorate_links2, dim3(nblk(maxl, 256), 4), dim3(256), 0, st, la, c->h1.as<double>(),
                         c->eri4.as<double>(), norb);
    }
  }
  for (int s = 0; s < 2; ++s) {
    SpinTables& t = c->sp[s];
    if (s == 0) {
      // merged same-spin CSR (singles then doubles of each row) for the row role's AXPY work items
      SQD_TRY(t.hs_ptr.reserve((t.n + 1) * 8));
      SQD_TRY(t.hs_src.reserve((size_t)(t.n_s + t.n_d) * 4));
      SQD_TRY(t.hs_val.reserve((size_t)(t.n_s + t.n_d) * 8));
      hipLaunchKernelGGL(k_merge_hs, dim3(nblk(t.n + 1, 4)), dim3(256), 0, st, t.n, t.s_ptr.as<int64_t>(),
                         t.d_ptr.as<int64_t>(), t.s_rec.as<SRec>(), t.s_val.as<double>(), t.d_src.as<uint32_t>(),
                         t.d_val.as<double>(), t.hs_ptr.as<int64_t>(), t.hs_src.as<uint32_t>(), t.hs_val.as<double>());
    }
    SQD_HIP_CHECK(hipGetLastError());
  }
  // capped sliced-ELL copies for the column role (beta): descriptors on the host, fill on the device
  {
    SpinTables& t = c->sp[1];
    VRowsHost& vs = c->hv_s;  // context members: they must outlive the asynchronous uploads
    VRowsHost& vd = c->hv_d;
    int cap = 8;
    if (const char* env = std::getenv("SQD_ELL_CAP")) {  // test hook: force tiny rows
      const int v = std::atoi(env);
      if (v >= 1) cap = v;
    }
    // LDS plan of the sigma kernel: K staged C rows + K integral rows + the row partial sums of one
    // column chunk.  Rows too long for that (nb beyond ~14 000) are not staged: the kernel then reads
    // them from global memory (L2) and works on column chunks, whose partial sums may use the freed LDS.
    const int cap0 = cap;
    const size_t budget = (size_t)c->lds_bytes - 8 * 1024;
    const size_t w2_bytes = (size_t)((nnorb + 1) & ~1) * 8;
    const size_t row_bytes = (size_t)((nb + 1) & ~int64_t(1)) * 8 + w2_bytes;
    bool lds_rows = (row_bytes + 64 <= budget) && ((nb + 1023) / 1024 <= 16);
    int64_t chunk_cols = nb;
    int64_t forced = 0;
    if (const char* env = std::getenv("SQD_SIGMA_GLOBAL_ROWS")) {  // test hook: chunk width, forces the fallback
      forced = (std::atoll(env) / 64) * 64;
      if (forced >= 64 && forced <= 1024) lds_rows = false;
      else forced = 0;
    }
    int64_t forced_pass = 0;  // test hook: partial-sum capacity in virtual rows, forces the multi-pass walk
    if (const char* env = std::getenv("SQD_SIGMA_PASS")) {
      forced_pass = std::atoll(env);
      if (forced_pass < 1 || !lds_rows) forced_pass = 0;
    }
    int64_t pass_s = 0, pass_d = 0;
    for (;;) {
      if (!lds_rows) chunk_cols = forced ? forced : 4096;
      const size_t target = lds_rows ? 40 * 1024 : 120 * 1024;
      for (cap = cap0;; cap *= 2) {
        make_vrows(c->h_sptr_b, nb, cap, chunk_cols, vs);
        make_vrows(c->h_dptr_b, nb, cap, chunk_cols, vd);
        if ((size_t)(vs.nv_max + vd.nv_max) * 8 <= target || cap >= (1 << 20)) break;
      }
      const size_t part_bytes = (size_t)(vs.nv_max + vd.nv_max) * 8 + 64;
      pass_s = vs.nv_max;
      pass_d = vd.nv_max;
      if (lds_rows && (row_bytes + part_bytes > budget || forced_pass)) {
        // The row fits but one partial sum per virtual row does not (at least one row per beta string with
        // links: nb of ~8000 and more).  Keep the row in LDS -- a gather from LDS beats a gather from L2
        // by far -- and walk the virtual rows in passes over a bounded partial-sum buffer.
        size_t avail = (budget > row_bytes + 128) ? budget - row_bytes - 128 : 0;
        if (std::getenv("SQD_SIGMA_NOPASS")) avail = 0;  // tuning hook: previous behaviour (global rows)
        if (avail >= 24 * 1024 || forced_pass) {
          cap = cap0 < 32 && !forced_pass ? 32 : cap0;  // moderate rows: balance without a row per 8 links
          make_vrows(c->h_sptr_b, nb, cap, chunk_cols, vs);
          make_vrows(c->h_dptr_b, nb, cap, chunk_cols, vd);
          const int64_t entries = (int64_t)(avail / 8);
          pass_s = forced_pass ? forced_pass : entries / 4;
          if (pass_s > vs.nv_max) pass_s = vs.nv_max;
          if (pass_s < 1) pass_s = 1;
          pass_d = forced_pass ? forced_pass : entries - ((pass_s + 1) & ~int64_t(1));
          if (!forced_pass && pass_d >= 1024) pass_d &= ~int64_t(1023);  // whole strides of the workgroup
          if (pass_d > vd.nv_max) pass_d = vd.nv_max;
          if (pass_d < 1) pass_d = 1;
          break;
        }
        lds_rows = false;
        continue;
      }
      if (!lds_rows && w2_bytes + part_bytes > budget) {
        set_error("beta link lists of a " + std::to_string(chunk_cols) + "-column chunk exceed the LDS budget");
        return SQD_ERR_LIMIT;
      }
      break;
    }
    c->sig_lds_rows = lds_rows;
    c->sig_ps = pass_s;
    c->sig_pd = pass_d;
    c->sig_chunk = chunk_cols;
    c->sig_nchunks = (int)((nb + chunk_cols - 1) / chunk_cols);
    c->sig_kmax = 4;
    // The kernel lives on resident waves: a single workgroup per CU (rows of ~2000 strings and more) is
    // the one case where a smaller batch pays (HF-centred 2000 x 2000: 2.3 ms -> 1.5 ms per sigma with 3
    // links per batch instead of 4).  Finer searches over cap / batch size to gain a third or fourth
    // resident workgroup were measured too and are a wash (+-10 % either way, profiles/r01 notes).
    if (lds_rows)
      while (c->sig_kmax > 2 && plan_sigma(c, nb, vs, vd, c->sig_kmax).wgs < 2) --c->sig_kmax;
    t.cap = cap;
    t.nv_s = vs.nv;
    t.nv_d = vd.nv;
    // the sigma work list is cut on the host from the same pointer arrays (no device dependency)
    c->na = na;
    c->nb = nb;
    c->D = na * nb;
    c->nelec[0] = nocc[0];
    c->nelec[1] = nocc[1];
    SQD_TRY(build_sigma_work(c));
    // all descriptors and the work list travel in ONE host blob / ONE copy; the per-array DevBufs are views into it
    struct Up { DevBuf* buf; const void* src; size_t bytes; };
    const Up ups[] = {
        {&t.vs_cnt, vs.vcnt.data(), vs.vcnt.size() * 4},   {&t.vs_own, vs.own.data(), vs.own.size() * 4},
        {&t.vs_start, vs.vstart.data(), vs.vstart.size() * 8}, {&t.es_sl, vs.sl.data(), vs.sl.size() * 8},
        {&t.vd_cnt, vd.vcnt.data(), vd.vcnt.size() * 4},   {&t.vd_own, vd.own.data(), vd.own.size() * 4},
        {&t.vd_start, vd.vstart.data(), vd.vstart.size() * 8}, {&t.ed_sl, vd.sl.data(), vd.sl.size() * 8},
        {&t.vs_chunk, vs.chunk.data(), vs.chunk.size() * 4}, {&t.vd_chunk, vd.chunk.data(), vd.chunk.size() * 4},
        {&c->items, c->h_items.data(), c->h_items.size() * sizeof(WorkItem)},
        {&c->multi, c->h_multi.data(), c->h_multi.size() * sizeof(MultiRow)},
    };
    size_t blob = 0;
    for (const Up& u : ups) blob += (u.bytes + 15) & ~size_t(15);
    void* h_blob = nullptr;
    SQD_TRY(stage_alloc(c, blob + 16, &h_blob));
    SQD_TRY(c->d_blob.reserve(blob + 16));
    size_t off = 0;
    for (const Up& u : ups) {
      if (u.bytes) std::memcpy(static_cast<char*>(h_blob) + off, u.src, u.bytes);
      u.buf->set_view(static_cast<char*>(c->d_blob.p) + off);
      off += (u.bytes + 15) & ~size_t(15);
    }
    if (blob) SQD_HIP_CHECK(hipMemcpyAsync(c->d_blob.p, h_blob, blob, hipMemcpyHostToDevice, st));
    SQD_TRY(t.es_rec.reserve((size_t)vs.total * sizeof(SRec) + 8));
    SQD_TRY(t.es_val.reserve((size_t)vs.total * 8 + 8));
    SQD_TRY(t.ed_src.reserve((size_t)vd.total * 4 + 8));
    SQD_TRY(t.ed_val.reserve((size_t)vd.total * 8 + 8));
    if (t.n_s > 0)
      hipLaunchKernelGGL(k_fill_vell_singles, dim3(nblk(vs.nv, 256)), dim3(256), 0, st, vs.nv,
                         (const int32_t*)t.vs_cnt.as<int32_t>(), (const int64_t*)t.vs_start.as<int64_t>(),
                         (const int64_t*)t.es_sl.as<int64_t>(), (const SRec*)t.s_rec.as<SRec>(),
                         (const double*)t.s_val.as<double>(), t.es_rec.as<SRec>(), t.es_val.as<double>());
    if (t.n_d > 0)
      hipLaunchKernelGGL(k_fill_vell_doubles, dim3(nblk(vd.nv, 256)), dim3(256), 0, st, vd.nv,
                         (const int32_t*)t.vd_cnt.as<int32_t>(), (const int64_t*)t.vd_start.as<int64_t>(),
                         (const int64_t*)t.ed_sl.as<int64_t>(), (const uint32_t*)t.d_src.as<uint32_t>(),
                         (const double*)t.d_val.as<double>(), t.ed_src.as<uint32_t>(), t.ed_val.as<double>());
    SQD_HIP_CHECK(hipGetLastError());
  }
  // no synchronisation here: later calls use the same stream; ev[0]..ev[1] is read lazily
  SQD_HIP_CHECK(hipEventRecord(c->ev[1], st));
  c->stage_pending = true;
  c->ms_setup = -1.0;
  c->have_subspace = true;
  return SQD_OK;
}

}  // namespace sqd
