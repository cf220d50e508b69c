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
#include "sqd_device.h"
#include "sqd_davstate.h"

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
                   &d_val, &jd_src, &jd_val, &hs_ptr, &hs_src, &hs_val, &jrow, &jT, &es_sl, &ed_sl, &es_rec, &es_val, &ed_src, &ed_val,
                   &vs_cnt, &vs_own, &vs_start, &vd_cnt, &vd_own, &vd_start, &vs_chunk, &vd_chunk};
  for (DevBuf* b : all) b->release();
}

// ---- pinned staging arena (see sqd_ctx::stage_*)
static int stage_reset(sqd_ctx* c) {
  if (c->stage_pending) {  // copies of the previous set_subspace may still read the arena (no full sync since)
    SQD_STREAM_SYNC(c->stream);
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
                           double* __restrict__ jm, double* __restrict__ km, double* __restrict__ jdiag) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n4 = (int64_t)norb * norb * norb * norb;
  if (idx >= n4) return;
  const int s = idx % norb;
  const int r = (idx / norb) % norb;
  const int q = (idx / ((int64_t)norb * norb)) % norb;
  const int p = idx / ((int64_t)norb * norb * norb);
  const double v = eri4[idx];
  if (p >= q && r >= s) eri_pp[(int64_t)tril(p, q) * nnorb + tril(r, s)] = v;
  if (p >= q && r == s) jdiag[(int64_t)tril(p, q) * norb + r] = v;  // (pq|kk), contiguous in k: the J tables' operand
  if (p == q && r == s) jm[p * norb + r] = v;
  if (p == s && q == r) km[p * norb + q] = v;
}

// ------------------------------------------------------------------ per-string tables
// e_str[I] = sum_{i in I} h_ii + 1/2 sum_{i,j in I} (J_ij - K_ij)
__device__ inline void string_energy_body(const uint64_t* __restrict__ strs, int64_t n, const double* __restrict__ h1,
                                          const double* __restrict__ jm, const double* __restrict__ km, int norb,
                                          double* __restrict__ e_str, unsigned bx) {
  // one wavefront per string: lane l takes the orbital pairs (i, j) = (l / nocc, l % nocc), l += 64;
  // the shuffle tree adds them in fixed order
  const int lane = threadIdx.x & 63;
  const int64_t I = (int64_t)bx * (blockDim.x >> 6) + (threadIdx.x >> 6);
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
__device__ inline void jtable_body(const uint64_t* __restrict__ strs, int64_t n, const double* __restrict__ jdiag,
                                   int nnorb, int norb, int transposed, double* __restrict__ out, unsigned bx) {
  const int64_t idx = (int64_t)bx * blockDim.x + threadIdx.x;
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
    v += jdiag[pair * norb + k];  // (the occupied orbitals of one string read 2-4 cache lines of one 8 * norb byte row)
  }
  out[idx] = v;
}

// ---- set_subspace is FOUR launches (round 1: nine).  At the sizes of one subsample batch every one of these
// kernels runs 3-6 us, less than it costs the host to enqueue it, so the work is grouped by dependency level and the
// jobs of one level share a launch through blockIdx.y:
//   A  k_tables_count : link counts of both spins, per-string mean-field energies of both spins; the workgroup that
//                       arrives last turns the counts into the four CSR pointer arrays (exclusive scans)
//      -> one D2H copy of the pointers (the host cuts the sigma work list and the capped-ELL descriptors from them)
//   B  k_tables_diag  : J tables of both spins, the diagonal (+ per-row minima for pyscf's init guess); needs only
//                       the strings and the energies, so it runs while the host waits for the copy and cuts the lists
//   C  k_tables_fill  : link enumeration into the CSR arrays, each link decorated (orbitals, sign, pair index,
//                       integral value) by the lane that found it
//   D  k_tables_ell   : merged same-spin CSR (alpha) + the capped sliced-ELL copies (beta)
struct SpinLinkArgs {
  GPtr<const uint64_t> strs;
  int64_t n, n_s, n_d;
  GPtr<int64_t> cnt_s, cnt_d;  // pass 1
  GPtr<int64_t> s_ptr, d_ptr;  // pass 2
  GPtr<SRec> s_rec;
  GPtr<uint32_t> s_row, d_src, d_row, d_orb;
  GPtr<double> s_val, d_val;
  GPtr<double> e_str;
  GPtr<double> jtab;  // jrow (alpha: [I][pair]) or jT (beta: [pair][I])
  int transposed;
};
struct SpinLinkArgs2 {
  SpinLinkArgs a[2];
};

// four independent exclusive scans by the LAST workgroup of k_tables_count, one wavefront each (256 threads = 4
// wavefronts; the first version ran them one after the other through a serial LDS prefix: 15 us).  Every pointer is
// written twice: to the device array the later kernels read and to its host-visible twin, so that the host -- which
// cuts the sigma work list and the ELL descriptors from them -- needs no copy command and no event.
struct ScanJobs {
  GPtr<const int64_t> in[4];
  GPtr<int64_t> out[4];
  int64_t n[4];
  GPtr<const int64_t> dev_block;  // the four pointer arrays, contiguous
  GPtr<int64_t> host_block;  // ... and their host-visible twin
  int64_t nptr;
  GPtr<long long> seq_word;  // host-visible: written last
  long long seq;
};
__device__ inline void wave_exclusive_scan(const int64_t* in, int64_t* __restrict__ out, int64_t n) {
  const int lane = threadIdx.x & 63;
  const int64_t chunk = (n + 63) / 64;
  const int64_t lo = (int64_t)lane * chunk;
  const int64_t hi = (lo + chunk < n) ? lo + chunk : n;
  // (eight loads in flight per round: as a plain loop this was one L2 round trip per element, twice -- 10 us)
  int64_t s = 0;
  for (int64_t i0 = lo; i0 < hi; i0 += 8) {
    int64_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = coherent_load_i64(&in[i0 + u < hi ? i0 + u : i0]);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += (i0 + u < hi) ? v[u] : 0;
  }
  // inclusive prefix over the lanes (Hillis-Steele on shuffles), then exclusive
  int64_t incl = s;
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t o = (int64_t)__shfl((long long)incl, lane - off >= 0 ? lane - off : lane);
    if (lane >= off) incl += o;
  }
  int64_t run = incl - s;
  const int64_t total = (int64_t)__shfl((long long)incl, 63);
  for (int64_t i0 = lo; i0 < hi; i0 += 8) {
    int64_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = coherent_load_i64(&in[i0 + u < hi ? i0 + u : i0]);
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < hi) {
        coherent_store_i64(&out[i0 + u], run);
        run += v[u];
      }
  }
  if (lane == 0) coherent_store_i64(&out[n], total);
}

// A: by = spin (link counts) | 2 + spin (string energies).  (bx, by) = this workgroup's place in ITS problem's grid of
// (nbx, 4) workgroups: the whole launch for k_tables_count, one z-slice of it for the batched k_tables_count_b.
struct CountArgs {
  SpinLinkArgs2 p;
  GPtr<const double> h1, jm, km;
  int norb;
  ScanJobs jobs;
  GPtr<unsigned> counter;
  unsigned gx;  // workgroups along x of this problem (batched launches: gridDim.x is the largest of them)
};
__device__ inline void tables_count_body(const CountArgs& g, unsigned bx, unsigned by, unsigned nbx) {
  const SpinLinkArgs2& p = g.p;
  const ScanJobs& jobs = g.jobs;
  if (by >= 2) {
    const SpinLinkArgs& a = p.a[by & 1];
    string_energy_body(a.strs, a.n, g.h1, g.jm, g.km, g.norb, a.e_str, bx);
    return;
  }
  {
    const SpinLinkArgs& a = p.a[by];
    // one wavefront per target string I.  pc = popcount(I ^ J): 2 -> single, 4 -> double.
    const int lane = threadIdx.x & 63;
    const int64_t I = (int64_t)bx * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (I < a.n) {
      const uint64_t sI = a.strs[I];
      int64_t cs = 0, cd = 0;
      for (int64_t j0 = 0; j0 < a.n; j0 += 64) {
        const int64_t J = j0 + lane;
        int pc = 0;
        if (J < a.n) pc = __popcll(sI ^ a.strs[J]);
        cs += __popcll(__ballot(pc == 2));
        cd += __popcll(__ballot(pc == 4));
      }
      if (lane == 0) {
        coherent_store_i64(&a.cnt_s[I], cs);
        coherent_store_i64(&a.cnt_d[I], cd);
      }
    }
  }
  if (!arrive_last(g.counter, by * nbx + bx, 2 * nbx)) return;
  {
    const int j = threadIdx.x >> 6;  // blockDim.x == 256: one wavefront per scan
    wave_exclusive_scan(jobs.in[j], jobs.out[j], jobs.n[j]);
  }
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  // the whole pointer block to its host-visible twin with unit-stride stores (the first version let every lane of the
  // scans write its own elements: ~800 scattered 8-byte PCIe writes, 25 us; consecutive lanes -> consecutive words
  // leave the chip as a few dozen full-line writes)
  for (int64_t i = threadIdx.x; i < jobs.nptr; i += blockDim.x)
    __hip_atomic_store(&jobs.host_block[i], coherent_load_i64(&jobs.dev_block[i]), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();
    *static_cast<volatile long long*>(static_cast<long long*>(jobs.seq_word)) = jobs.seq;
  }
}
__global__ void k_tables_count(const CountArgs g) { tables_count_body(g, blockIdx.x, blockIdx.y, gridDim.x); }
// batched: blockIdx.z = problem; the arguments of every problem live in device memory (uniform address: scalar loads)
__global__ void k_tables_count_b(const CountArgs* __restrict__ gs) {
  const CountArgs& g = gs[blockIdx.z];  // (by reference: the bodies index into the record dynamically)
  if (blockIdx.x >= g.gx) return;
  tables_count_body(g, blockIdx.x, blockIdx.y, g.gx);
}

// B: by = 0 alpha J table | 1 beta J table (transposed) | 2 the diagonal, one alpha string per workgroup:
//   hdiag[A,B] = e_a[A] + e_b[B] + sum_{k in B} v_A[k],  v_A[k] = sum_{i in A} (ii|kk)   (pyscf make_hdiag)
// and the row's lowest element (over B <= A when tril_only: pyscf's init-guess rule for equal spin sectors)
struct DiagArgs {
  SpinLinkArgs2 p;
  GPtr<const double> jm, jdiag;
  int norb, nnorb;
  int64_t row0, row1, nb;
  int tril_only;
  GPtr<double> hdiag, pmin;
  GPtr<int64_t> pidx;
  unsigned gx;
  int coherent_min;  // the row minima are read by another workgroup of the SAME launch (k_tables_diag_fill)
};
__device__ inline void tables_diag_body(const DiagArgs& g, unsigned bx, unsigned by, unsigned nbx) {
  __shared__ double v[SQD_MAX_NORB];
  const SpinLinkArgs2& p = g.p;
  const int norb = g.norb;
  const int64_t row0 = g.row0, nb = g.nb;
  if (by < 2) {
    const SpinLinkArgs& a = p.a[by];
    jtable_body(a.strs, a.n, g.jdiag, g.nnorb, norb, a.transposed, a.jtab, bx);
    return;
  }
  for (int64_t A = row0 + bx; A < g.row1; A += nbx) {
    const uint64_t sA = p.a[0].strs[A];
    __syncthreads();
    if ((int)threadIdx.x < norb) {
      uint64_t occ = sA;
      double t = 0.0;
      while (occ) {
        const int i = ctz64(occ);
        occ &= occ - 1;
        t += g.jm[i * norb + threadIdx.x];
      }
      v[threadIdx.x] = t;
    }
    __syncthreads();
    const double ea = p.a[0].e_str[A];
    double best = 1e300;
    int64_t bi = -1;
    for (int64_t B = threadIdx.x; B < nb; B += blockDim.x) {
      uint64_t occ = p.a[1].strs[B];
      double t = ea + p.a[1].e_str[B];
      while (occ) {
        const int k = ctz64(occ);
        occ &= occ - 1;
        t += v[k];
      }
      g.hdiag[(A - row0) * nb + B] = t;
      if (!(g.tril_only && A < B) && (t < best || bi < 0)) {  // B ascends: the first minimum wins
        best = t;
        bi = (A - row0) * nb + B;
      }
    }
    block_argmin(best, bi);
    if (threadIdx.x == 0) {
      if (g.coherent_min) {
        coherent_store(&g.pmin[A - row0], best);
        coherent_store_i64(&g.pidx[A - row0], bi);
      } else {
        g.pmin[A - row0] = best;
        g.pidx[A - row0] = bi;
      }
    }
  }
}
__global__ void k_tables_diag(const DiagArgs g) { tables_diag_body(g, blockIdx.x, blockIdx.y, gridDim.x); }
__global__ void k_tables_diag_b(const DiagArgs* __restrict__ gs) {
  const DiagArgs& g = gs[blockIdx.z];  // (by reference: the bodies index into the record dynamically)
  if (blockIdx.x >= g.gx) return;
  tables_diag_body(g, blockIdx.x, blockIdx.y, g.gx);
}

// C: one wavefront per target string (blockIdx.y = spin): enumerate, compact with ballot + prefix popcount, and let
// the lane that found a link decorate it
// The launch also prepares the Davidson run that normally follows (job.x != nullptr): state block, arrival counters and
// pyscf's start vector from the row minima k_tables_diag left -- the k_init_guess launch of the solver, folded in
// here because this is the last launch of the table build that every string set with any link goes through.
struct GuessJob {
  GPtr<double> x;  // X[0] of the Davidson workspace (nullptr: no job)
  GPtr<const double> pmin;
  GPtr<const int64_t> pidx;
  int nrows;
  int64_t n;  // D
  GPtr<DavState> st;
  GPtr<unsigned> counter;
};
__device__ inline void tables_fill_wave(const SpinLinkArgs& a, const double* __restrict__ h1,
                                        const double* __restrict__ eri4, int norb, unsigned bx);
struct FillArgs {
  SpinLinkArgs2 p;
  GPtr<const double> h1, eri4;
  int norb;
  GuessJob job;
  unsigned gx;
};
__device__ inline void tables_fill_body(const FillArgs& g, unsigned bx, unsigned by, unsigned nbx) {
  tables_fill_wave(g.p.a[by], g.h1, g.eri4, g.norb, bx);
  const GuessJob& job = g.job;
  if (!job.x) return;
  if (bx == 0 && by == 0) dav_state_init(job.st, job.counter);
  const int64_t blk = (int64_t)by * nbx + bx;
  init_guess_write(job.n, job.pmin, job.pidx, job.nrows, job.x, blk * blockDim.x + threadIdx.x,
                   (int64_t)nbx * 2 * blockDim.x);
}
__global__ void k_tables_fill(const FillArgs g) { tables_fill_body(g, blockIdx.x, blockIdx.y, gridDim.x); }
__global__ void k_tables_fill_b(const FillArgs* __restrict__ gs) {
  const FillArgs& g = gs[blockIdx.z];  // (by reference: the bodies index into the record dynamically)
  if (blockIdx.x >= g.gx) return;
  tables_fill_body(g, blockIdx.x, blockIdx.y, g.gx);
}
// B + C in ONE launch (single solves of batch size: both string lists <= 1024, link arrays reserved at their upper
// bound, so the launch needs nothing from the host's look at the CSR pointers and goes out right behind launch A; the
// headline solve's table build is then count -> [diag | fill] with no host wait in between: 12.3 + 8.6 + 3.9 (gap) +
// 8.7 us -> 12.3 + ~9).  by = 0, 1: J tables; 2: diagonal + row minima; 3, 4: link fill of spin by - 3.
// The Davidson start vector needs the row minima of THIS launch: the fill workgroups zero X[0] with write-through
// stores (a plain store would sit dirty in its XCD's L2 and could be written back over the value below), every diag and
// fill workgroup arrives on a counter, and the last one finishes the argmin and writes the (at most three) non-zero
// elements -- init_guess_write's values, bit for bit.
struct DiagFillArgs {
  DiagArgs d;
  FillArgs f;
  GPtr<unsigned> counter;  // an arrival-counter set of its own (dav_state_init resets the Davidson's)
};
__global__ void __launch_bounds__(256) k_tables_diag_fill(const DiagFillArgs g) {
  const unsigned by = blockIdx.y, bx = blockIdx.x;
  const GuessJob& job = g.f.job;
  unsigned me;
  if (by < 3) {
    if (bx >= g.d.gx) return;
    tables_diag_body(g.d, bx, by, g.d.gx);
    if (by < 2) return;
    me = bx;
  } else {
    if (bx >= g.f.gx) return;
    tables_fill_wave(g.f.p.a[by - 3], g.f.h1, g.f.eri4, g.f.norb, bx);
    if (bx == 0 && by == 3) dav_state_init(job.st, job.counter);
    const int64_t blk = (int64_t)(by - 3) * g.f.gx + bx;
    for (int64_t i = blk * blockDim.x + threadIdx.x; i < job.n; i += (int64_t)g.f.gx * 2 * blockDim.x)
      coherent_store(&job.x[i], 0.0);
    me = g.d.gx + (by - 3) * g.f.gx + bx;
  }
  if (!arrive_last(g.counter, me, g.d.gx + 2 * g.f.gx)) return;
  double best = 1e300;
  int64_t bi = -1;
  for (int b = threadIdx.x; b < job.nrows; b += blockDim.x) {
    const double v = coherent_load(&job.pmin[b]);
    const int64_t i = coherent_load_i64(&job.pidx[b]);
    if (i >= 0 && (v < best || (v == best && i < bi) || bi < 0)) {
      best = v;
      bi = i;
    }
  }
  block_argmin(best, bi);
  if (threadIdx.x == 0) {
    const int64_t n = job.n, addr = bi < 0 ? 0 : bi;
    auto f = [=](int64_t i) { return ((i == addr) ? 1.0 : 0.0) + ((i == 0) ? 1e-5 : 0.0) - ((i == n - 1) ? 1e-5 : 0.0); };
    double nn = f(0) * f(0);
    if (n - 1 != 0) nn += f(n - 1) * f(n - 1);
    if (addr != 0 && addr != n - 1) nn += f(addr) * f(addr);
    const double inv = 1.0 / sqrt(nn);
    coherent_store(&job.x[0], f(0) * inv);
    coherent_store(&job.x[n - 1], f(n - 1) * inv);
    coherent_store(&job.x[addr], f(addr) * inv);
  }
}

__device__ inline void tables_fill_wave(const SpinLinkArgs& a, const double* __restrict__ h1,
                                        const double* __restrict__ eri4, int norb, unsigned bx) {
  const uint64_t* __restrict__ strs = a.strs;
  const int64_t n = a.n;
  const int lane = threadIdx.x & 63;
  const int64_t I = (int64_t)bx * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (I >= n) return;
  const uint64_t sI = strs[I];
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int64_t n1 = norb, n2 = n1 * norb, n3 = n2 * norb;
  int64_t ps = a.s_ptr[I], pd = a.d_ptr[I];
  for (int64_t j0 = 0; j0 < n; j0 += 64) {
    const int64_t J = j0 + lane;
    uint64_t sJ = 0;
    int pc = 0;
    if (J < n) {
      sJ = strs[J];
      pc = __popcll(sI ^ sJ);
    }
    const unsigned long long ms = __ballot(pc == 2);
    const unsigned long long md = __ballot(pc == 4);
    if (pc == 2) {
      // |I> = sign a+_cre a_des |J>;  value = sign * (h[cre,des] + sum_{k in J, k != des} (cre des|kk) - (cre k|k des))
      const int64_t pos = ps + __popcll(ms & lt);
      const uint64_t x = sI ^ sJ;
      const int ca = ctz64(x & sI), cb = ctz64(x & sJ);
      const int lo = ca < cb ? ca : cb, hi = ca < cb ? cb : ca;
      const uint64_t between = below_mask(hi) & ~below_mask(lo + 1);
      const int neg = __popcll(sJ & between) & 1;
      double val = h1[ca * norb + cb];
      uint64_t occ = sJ & ~(1ull << cb);
      while (occ) {
        const int k = ctz64(occ);
        occ &= occ - 1;
        val += eri4[ca * n3 + cb * n2 + k * n1 + k] - eri4[ca * n3 + k * n2 + k * n1 + cb];
      }
      const uint32_t widx = 2u * tril(ca, cb) + (ca > cb ? 1u : 0u);
      a.s_rec[pos] = SRec{(uint32_t)J, widx | ((uint32_t)ca << 13) | ((uint32_t)cb << 19) | ((uint32_t)neg << 31)};
      a.s_row[pos] = (uint32_t)I;
      a.s_val[pos] = neg ? -val : val;
    }
    if (pc == 4) {
      // |I> = sign a+_p a+_r a_s a_q |J>, p>r, q>s;  value = sign * ((pq|rs) - (ps|rq))
      const int64_t pos = pd + __popcll(md & lt);
      const uint64_t x = sI ^ sJ;
      uint64_t cre = x & sI, des = x & sJ;
      const int r = ctz64(cre);
      cre &= cre - 1;
      const int pp = ctz64(cre);
      const int s = ctz64(des);
      des &= des - 1;
      const int q = ctz64(des);
      // apply a_q, a_s, a+_r, a+_p in that order, collecting parities
      uint64_t st = sJ;
      int par = __popcll(st & below_mask(q));
      st ^= 1ull << q;
      par += __popcll(st & below_mask(s));
      st ^= 1ull << s;
      par += __popcll(st & below_mask(r));
      st |= 1ull << r;
      par += __popcll(st & below_mask(pp));
      const int neg = par & 1;
      const double val = eri4[pp * n3 + q * n2 + r * n1 + s] - eri4[pp * n3 + s * n2 + r * n1 + q];
      a.d_src[pos] = (uint32_t)J;
      a.d_row[pos] = (uint32_t)I;
      a.d_val[pos] = neg ? -val : val;
      a.d_orb[pos] = (uint32_t)pp | ((uint32_t)r << 6) | ((uint32_t)q << 12) | ((uint32_t)s << 18) | ((uint32_t)neg << 31);
    }
    ps += __popcll(ms);
    pd += __popcll(md);
  }
}

// The double links of 64 consecutive strings (a slice) re-ordered for k_sigma_rows: step k of every list first, lists
// shorter than k skipped -- the slice keeps its CSR range [d_ptr[64 s], d_ptr[64 s + 64]) and needs no index of its
// own: a reader rebuilds positions from ballots exactly as this writer does.  One wavefront per slice.
struct JdsArgs {
  int64_t n;
  GPtr<const int64_t> d_ptr;
  GPtr<const uint32_t> d_src;
  GPtr<const double> d_val;
  GPtr<uint32_t> jd_src;
  GPtr<double> jd_val;
  unsigned gx;
};
__device__ inline void tables_jds_body(int64_t n, const int64_t* __restrict__ d_ptr, const uint32_t* __restrict__ d_src,
                                       const double* __restrict__ d_val, uint32_t* __restrict__ jd_src,
                                       double* __restrict__ jd_val, unsigned bx) {
  const int lane = threadIdx.x & 63;
  const int64_t B0 = ((int64_t)bx * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
  if (B0 >= n) return;
  const int64_t B = B0 + lane;
  const int64_t d0 = (B < n) ? d_ptr[B] : 0;
  const int len = (B < n) ? (int)(d_ptr[B + 1] - d0) : 0;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int64_t base = d_ptr[B0];
  for (int k = 0;; ++k) {
    const unsigned long long m = __ballot(k < len);
    if (!m) break;
    if (k < len) {
      const int64_t pos = base + __popcll(m & lt);
      jd_src[pos] = d_src[d0 + k];
      jd_val[pos] = d_val[d0 + k];
    }
    base += __popcll(m);
  }
}
__global__ void k_tables_jds(const JdsArgs g) {
  tables_jds_body(g.n, g.d_ptr, g.d_src, g.d_val, g.jd_src, g.jd_val, blockIdx.x);
}
__global__ void k_tables_jds_b(const JdsArgs* __restrict__ gs) {
  const JdsArgs& g = gs[blockIdx.z];  // (by reference: the bodies index into the record dynamically)
  if (blockIdx.x >= g.gx) return;
  tables_jds_body(g.n, g.d_ptr, g.d_src, g.d_val, g.jd_src, g.jd_val, blockIdx.x);
}

// Dense, zero-padded copy of one spin's same-spin block (singles' same-spin value + doubles), row `bx`: the operand of
// k_same_spin_mfma.  The block is symmetric and every row's CSR lists hold all of its sources, so a workgroup writes
// its row alone: zero, barrier, scatter.  by = spin.
struct DenseFillArgs {
  int64_t n[2];
  int P[2];
  GPtr<const int64_t> s_ptr[2], d_ptr[2];
  GPtr<const SRec> s_rec[2];
  GPtr<const double> s_val[2], d_val[2];
  GPtr<const uint32_t> d_src[2];
  GPtr<double> out[2];
  unsigned gx;
};
__device__ inline void tables_dense_body(const DenseFillArgs& g, unsigned bx, unsigned by) {
  const int P = g.P[by];
  if ((int)bx >= P) return;
  double* __restrict__ row = g.out[by] + (int64_t)bx * P;
  for (int j = threadIdx.x; j < P; j += blockDim.x) row[j] = 0.0;
  __syncthreads();
  if ((int64_t)bx >= g.n[by]) return;
  const int64_t s0 = g.s_ptr[by][bx], s1 = g.s_ptr[by][bx + 1], d0 = g.d_ptr[by][bx], d1 = g.d_ptr[by][bx + 1];
  for (int64_t l = s0 + threadIdx.x; l < s1; l += blockDim.x) row[g.s_rec[by][l].src] = g.s_val[by][l];
  for (int64_t l = d0 + threadIdx.x; l < d1; l += blockDim.x) row[g.d_src[by][l]] = g.d_val[by][l];
}
__global__ void k_tables_dense(const DenseFillArgs g) { tables_dense_body(g, blockIdx.x, blockIdx.y); }
__global__ void k_tables_dense_b(const DenseFillArgs* __restrict__ gs) {
  const DenseFillArgs& g = gs[blockIdx.z];  // (by reference: indexed dynamically)
  if (blockIdx.x >= g.gx) return;
  tables_dense_body(g, blockIdx.x, blockIdx.y);
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
struct EllArgs {
  // merged same-spin CSR of the row role (alpha): row i = its single links (value incl. sign) then its double links
  int64_t n_a;
  GPtr<const int64_t> sa_ptr, da_ptr;
  GPtr<const SRec> sa_rec;
  GPtr<const double> sa_val;
  GPtr<const uint32_t> da_src;
  GPtr<const double> da_val;
  GPtr<int64_t> hs_ptr;
  GPtr<uint32_t> hs_src;
  GPtr<double> hs_val;
  // capped sliced-ELL copies of the column role (beta)
  int64_t nv_s, nv_d;
  GPtr<const int32_t> vs_cnt, vd_cnt;
  GPtr<const int64_t> vs_start, vd_start, es_sl, ed_sl;
  GPtr<const SRec> sb_rec;
  GPtr<const double> sb_val;
  GPtr<const uint32_t> db_src;
  GPtr<const double> db_val;
  GPtr<SRec> es_rec;
  GPtr<double> es_val;
  GPtr<uint32_t> ed_src;
  GPtr<double> ed_val;
  unsigned gx;
};
// D: blockIdx.y = 0 merged alpha CSR (one wavefront per row: rows of the Hartree-Fock neighbourhood hold hundreds
// of links) | 1 beta singles' ELL (thread per virtual row) | 2 beta doubles' ELL
__device__ inline void tables_ell_body(const EllArgs& g, unsigned bx, unsigned by) {
  if (by == 0) {
    const int64_t i = (int64_t)bx * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i > g.n_a) return;
    const int64_t o = g.sa_ptr[i] + g.da_ptr[i];
    if (lane == 0) g.hs_ptr[i] = o;
    if (i == g.n_a) return;
    const int64_t s0 = g.sa_ptr[i], ns = g.sa_ptr[i + 1] - s0, d0 = g.da_ptr[i], nd = g.da_ptr[i + 1] - d0;
    for (int64_t k = lane; k < ns; k += 64) {
      g.hs_src[o + k] = g.sa_rec[s0 + k].src;
      g.hs_val[o + k] = g.sa_val[s0 + k];
    }
    for (int64_t k = lane; k < nd; k += 64) {
      g.hs_src[o + ns + k] = g.da_src[d0 + k];
      g.hs_val[o + ns + k] = g.da_val[d0 + k];
    }
    return;
  }
  const int64_t v = (int64_t)bx * blockDim.x + threadIdx.x;
  if (by == 1) {
    if (v >= g.nv_s) return;
    const int64_t base = g.es_sl[v >> 6] + (v & 63);
    const int64_t p0 = g.vs_start[v];
    const int cnt = g.vs_cnt[v];
    for (int k = 0; k < cnt; ++k) {
      g.es_rec[base + (int64_t)k * 64] = g.sb_rec[p0 + k];
      g.es_val[base + (int64_t)k * 64] = g.sb_val[p0 + k];
    }
  } else {
    if (v >= g.nv_d) return;
    const int64_t base = g.ed_sl[v >> 6] + (v & 63);
    const int64_t p0 = g.vd_start[v];
    const int cnt = g.vd_cnt[v];
    for (int k = 0; k < cnt; ++k) {
      g.ed_src[base + (int64_t)k * 64] = g.db_src[p0 + k];
      g.ed_val[base + (int64_t)k * 64] = g.db_val[p0 + k];
    }
  }
}
__global__ void k_tables_ell(const EllArgs g) { tables_ell_body(g, blockIdx.x, blockIdx.y); }
__global__ void k_tables_ell_b(const EllArgs* __restrict__ gs) {
  const EllArgs& g = gs[blockIdx.z];  // (by reference: the bodies index into the record dynamically)
  if (blockIdx.x >= g.gx) return;
  tables_ell_body(g, blockIdx.x, blockIdx.y);
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
  // nb = 2048, 1024 beyond; never more than the row or the virtual-row lists can occupy.  Round 3, rows of up to 384
  // strings: 256 threads with two columns and two virtual rows each -- twice the workgroups in flight per CU for items
  // whose time is their chain of dependent loads (16 x HF-centred 317^2: 1.20 -> 1.09 ms per batch, single solve 3.00 ->
  // 2.96; the loop's 200-300-string subspaces 5.6 -> 5.2 and 7.9 -> 7.5 ms per iteration; equal from ~430 strings on and
  // worse at 707: 6.4 -> 7.3 ms)
  const int64_t nvmax = vs.nv > vd.nv ? vs.nv : vd.nv;
  const int64_t want = nb > nvmax ? nb : nvmax;
  int T = (int)(((want + 63) / 64) * 64);
  const int tmax = (nb <= 384) ? 256 : (nb <= 2048) ? 512 : 1024;
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
  c->h_rowinfo.assign((size_t)2 * (c->row1 - c->row0), 0);
  int32_t nslots = 0;
  std::vector<WorkItem> row;
  for (int64_t A = c->row0; A < c->row1; ++A) {
    row.clear();
    const int64_t s0 = c->h_sptr[A], s1 = c->h_sptr[A + 1];
    const int64_t h0 = s0 + c->h_dptr[A], h1 = s1 + c->h_dptr[A + 1];
    // (dense same-spin blocks: the alpha links are in the matrix-core product, no AXPY work in any item)
    WorkItem own{h0, (uint32_t)A, 0, (uint16_t)(c->sig_dense ? 0 : ((h1 - h0 < L0) ? (h1 - h0) : L0)), -1, 0};
    row.push_back(own);
    for (int64_t l = s0; l < s1; l += K)
      row.push_back(WorkItem{l, (uint32_t)A, 1, (uint16_t)((s1 - l < K) ? (s1 - l) : K), -1, 0});
    if (!c->sig_dense)
      for (int64_t l = h0 + L0; l < h1; l += L)
        row.push_back(WorkItem{l, (uint32_t)A, 2, (uint16_t)((h1 - l < L) ? (h1 - l) : L), -1, 0});
    if (row.size() > 1) {  // several items: partial rows + fixed-order reduce
      multi.push_back(MultiRow{(uint32_t)A, nslots, (int32_t)row.size()});
      c->h_rowinfo[2 * (A - c->row0)] = nslots;
      c->h_rowinfo[2 * (A - c->row0) + 1] = (int32_t)row.size();
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
  SQD_TRY(c->jdiag.reserve((int64_t)nnorb * norb * 8));
  SQD_HIP_CHECK(hipMemcpyAsync(c->h1.p, h1, n2 * 8, hipMemcpyHostToDevice, c->stream));
  SQD_HIP_CHECK(hipMemcpyAsync(c->eri4.p, eri, n4 * 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_pack_eri, dim3(nblk(n4, 256)), dim3(256), 0, c->stream, c->eri4.as<double>(), norb, nnorb,
                     c->eri_pp.as<double>(), c->jm.as<double>(), c->km.as<double>(), c->jdiag.as<double>());
  SQD_HIP_CHECK(hipGetLastError());
  SQD_STREAM_SYNC(c->stream);
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

// ---- the table build of ONE subspace, cut into phases so that the same code serves a single solve (every phase
// launches its own kernels) and a batched solve (sqd_solve_batch: the phases of all subspaces run in lock-step and
// each level's kernels of all of them go out as ONE launch, arguments in device memory).
struct SubspaceBuild {
  const uint64_t *sa = nullptr, *sb = nullptr;
  int64_t na = 0, nb = 0, row0 = 0, row1 = 0, nrows = 0, maxn = 0;
  int nocc[2] = {0, 0};
  long long seq_ptrs = 0;
  int64_t tot[4] = {0, 0, 0, 0};
  CountArgs count;
  DiagArgs diag;
  FillArgs fill;
  EllArgs ell;
  JdsArgs jds;
  DenseFillArgs dense;
  bool have_fill = false, have_ell = false, have_jds = false, have_dense = false;
  bool fill_launched = false;  // launch C went out together with launch B, before the host saw the link counts
  std::vector<int64_t> zero_ptr;  // dense same-spin blocks: the beta doubles leave the work items (empty lists)
  unsigned ell_gx = 0;
  struct Up {
    DevBuf* buf;
    const void* src;
    size_t bytes;
  };
  std::vector<Up> ups;    // work-item sigma: descriptor arrays that travel in one blob
  size_t blob_bytes = 0;
};

// phase 1: checks, buffers, arguments of launches A (counts + scans) and B (J tables, diagonal).  `strs_dev`: where
// the na + nb strings will be in device memory (nullptr: the context's own buffer); the caller uploads them.
static int subspace_phase1(sqd_ctx* c, SubspaceBuild& b, const uint64_t* sa, int64_t na, const uint64_t* sb, int64_t nb,
                           int64_t row0, int64_t row1, uint64_t* strs_dev) {
  c->have_subspace = false;
  c->guess_x = nullptr;
  c->have_solution = false;
  if (row1 < 0) row1 = na;
  if (na <= 0 || sa == nullptr) {
    set_error("empty Spin-up string list");
    return SQD_ERR_INVALID;
  }
  if (row0 < 0 || row0 >= row1 || row1 > na) {
    set_error("alpha row range [row0, row1) must be non-empty and inside [0, na)");
    return SQD_ERR_INVALID;
  }
  b.sa = sa;
  b.sb = sb;
  b.na = na;
  b.nb = nb;
  b.row0 = row0;
  b.row1 = row1;
  b.nrows = row1 - row0;
  SQD_TRY(validate_strings(sa, na, c->norb, "Spin-up", &b.nocc[0]));
  SQD_TRY(validate_strings(sb, nb, c->norb, "Spin-down", &b.nocc[1]));
  const int norb = c->norb, nnorb = c->nnorb;
  const int64_t ns[2] = {na, nb};
  const int64_t maxn = na > nb ? na : nb;
  b.maxn = maxn;
  SQD_TRY(c->scratch.reserve((size_t)(4 * maxn + 4 * (maxn / 64 + 2) + 64) * 8));
  SQD_TRY(reserve_counters(c));
  // both string lists in one device buffer
  if (strs_dev) c->strs2.set_view(strs_dev);
  else SQD_TRY(c->strs2.reserve((size_t)(na + nb) * 8));
  for (int s = 0; s < 2; ++s) {
    SpinTables& t = c->sp[s];
    t.n = ns[s];
    t.nocc = b.nocc[s];
    t.n_slices = (t.n + 63) / 64;
    t.strs.set_view(c->strs2.as<uint64_t>() + (s ? na : 0));
    SQD_TRY(t.e_str.reserve(t.n * 8));
  }
  // launch A: counts + CSR pointers for both spins (+ per-string energies).  All four CSR pointer arrays live in one
  // device buffer with a host-visible twin: the sigma work list (alpha) and the capped-ELL geometry (beta) are cut on
  // the host from these pointers
  const int64_t nptr = 2 * (na + 1) + 2 * (nb + 1);
  SQD_TRY(c->ptrs.reserve((size_t)nptr * 8));
  if (c->ptrs_map_cap < (size_t)nptr) {  // host-visible twin of the pointer block (grow-only)
    if (c->h_ptrs_map) SQD_HIP_CHECK(hipHostFree(c->h_ptrs_map));
    c->h_ptrs_map = nullptr;
    c->ptrs_map_cap = 0;
    const size_t want = (size_t)nptr + (size_t)nptr / 4 + 64;
    SQD_HIP_CHECK(hipHostMalloc((void**)&c->h_ptrs_map, want * 8, hipHostMallocMapped | hipHostMallocCoherent));
    SQD_HIP_CHECK(hipHostGetDevicePointer((void**)&c->d_ptrs_map, c->h_ptrs_map, 0));
    c->ptrs_map_cap = want;
  }
  {
    int64_t* base = c->ptrs.as<int64_t>();
    c->sp[0].s_ptr.set_view(base);
    c->sp[0].d_ptr.set_view(base + (na + 1));
    c->sp[1].s_ptr.set_view(base + 2 * (na + 1));
    c->sp[1].d_ptr.set_view(base + 2 * (na + 1) + (nb + 1));
  }
  SpinLinkArgs2 la;  // both spins' arguments, filled in as the buffers come to exist
  std::memset(&la, 0, sizeof(la));
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
    la.a[s].e_str = t.e_str.as<double>();
    jobs.in[2 * s] = cnt_s;
    jobs.out[2 * s] = t.s_ptr.as<int64_t>();
    jobs.n[2 * s] = t.n;
    jobs.in[2 * s + 1] = cnt_d;
    jobs.out[2 * s + 1] = t.d_ptr.as<int64_t>();
    jobs.n[2 * s + 1] = t.n;
  }
  jobs.dev_block = c->ptrs.as<int64_t>();
  jobs.host_block = c->d_ptrs_map;
  jobs.nptr = nptr;
  jobs.seq = ++c->mail_seq;
  jobs.seq_word = reinterpret_cast<long long*>(c->d_mail + 3 * 128 + 256);  // its own word of the mailbox page
  b.seq_ptrs = jobs.seq;
  // launch B -- everything else that needs only the strings: occupation (J) tables, the diagonal, the row minima
  SQD_TRY(c->hdiag.reserve((size_t)b.nrows * nb * 8));
  SQD_TRY(c->guess_min.reserve((size_t)b.nrows * 16));
  for (int s = 0; s < 2; ++s) {
    SpinTables& t = c->sp[s];
    DevBuf& jt = (s == 0) ? t.jrow : t.jT;  // alpha: J[I][pair] (row role); beta: transposed (column role)
    SQD_TRY(jt.reserve(t.n * nnorb * 8));
    la.a[s].jtab = jt.as<double>();
    la.a[s].transposed = s;
  }
  CountArgs& ca = b.count;
  ca.p = la;
  ca.h1 = c->h1.as<double>();
  ca.jm = c->jm.as<double>();
  ca.km = c->km.as<double>();
  ca.norb = norb;
  ca.jobs = jobs;
  ca.counter = counter_ptr(c);
  ca.gx = nblk(maxn, 4);
  DiagArgs& da = b.diag;
  const int64_t gx_j = (int64_t)nblk(maxn * nnorb, 256), gx_h = b.nrows < 65535 ? b.nrows : 65535;
  double* pmin = c->guess_min.as<double>();
  da.p = la;
  da.jm = c->jm.as<double>();
  da.jdiag = c->jdiag.as<double>();
  da.norb = norb;
  da.nnorb = nnorb;
  da.row0 = row0;
  da.row1 = row1;
  da.nb = nb;
  da.tril_only = (b.nocc[0] == b.nocc[1] && na == nb) ? 1 : 0;
  da.hdiag = c->hdiag.as<double>();
  da.pmin = pmin;
  da.pidx = reinterpret_cast<int64_t*>(pmin + b.nrows);
  da.gx = (unsigned)(gx_j > gx_h ? gx_j : gx_h);
  da.coherent_min = 0;
  return SQD_OK;
}

// between the phases: the scan of launch A posted the pointers to their host-visible twin -- spin on its sequence word
static int subspace_wait_pointers(sqd_ctx* c, SubspaceBuild& b) {
  const int64_t na = b.na, nb = b.nb;
  SQD_TRY(spin_wait_word(c->h_mail + 3 * 128 + 256, b.seq_ptrs, c->stream));
  c->h_sptr = c->h_ptrs_map;
  c->h_dptr = c->h_sptr + (na + 1);
  c->h_sptr_b = c->h_dptr + (na + 1);
  c->h_dptr_b = c->h_sptr_b + (nb + 1);
  b.tot[0] = c->h_sptr[na];
  b.tot[1] = c->h_dptr[na];
  b.tot[2] = c->h_sptr_b[nb];
  b.tot[3] = c->h_dptr_b[nb];
  return SQD_OK;
}

// arguments of launch C (fill + decorate + the start of the Davidson run), link arrays reserved for cap[2 s] single and
// cap[2 s + 1] double links of spin s: the counted totals -- or, for the launch that goes out BEFORE the host has seen
// them, their upper bound
static int subspace_fill_setup(sqd_ctx* c, SubspaceBuild& b, const int64_t* cap, bool want_fill) {
  const int64_t na = b.na, nb = b.nb, row0 = b.row0, row1 = b.row1, nrows = b.nrows;
  SpinLinkArgs2 la = b.count.p;
  for (int s = 0; s < 2; ++s) {
    SpinTables& t = c->sp[s];
    SQD_TRY(t.s_rec.reserve((size_t)cap[2 * s] * sizeof(SRec)));
    SQD_TRY(t.s_row.reserve((size_t)cap[2 * s] * 4));
    SQD_TRY(t.s_val.reserve((size_t)cap[2 * s] * 8));
    SQD_TRY(t.d_src.reserve((size_t)cap[2 * s + 1] * 4));
    SQD_TRY(t.d_row.reserve((size_t)cap[2 * s + 1] * 4));
    SQD_TRY(t.d_orb.reserve((size_t)cap[2 * s + 1] * 4));
    SQD_TRY(t.d_val.reserve((size_t)cap[2 * s + 1] * 8));
    la.a[s].jtab = b.diag.p.a[s].jtab;
    la.a[s].transposed = s;
    la.a[s].n_s = cap[2 * s];
    la.a[s].n_d = cap[2 * s + 1];
    la.a[s].s_rec = t.s_rec.as<SRec>();
    la.a[s].s_row = t.s_row.as<uint32_t>();
    la.a[s].s_val = t.s_val.as<double>();
    la.a[s].d_src = t.d_src.as<uint32_t>();
    la.a[s].d_row = t.d_row.as<uint32_t>();
    la.a[s].d_orb = t.d_orb.as<uint32_t>();
    la.a[s].d_val = t.d_val.as<double>();
  }
  b.have_fill = false;
  if (want_fill) {
    GuessJob job;
    std::memset(&job, 0, sizeof(job));
    c->guess_x = nullptr;
    if (row0 == 0 && row1 == na && (size_t)c->dav_nvecs_hint * na * nb * 8 <= (size_t(1) << 30)) {
      // (small problems only: there a launch matters, and the workspace costs nothing to have early)
      // the Davidson workspace at the size the last run used (default max_space + 1 vectors): if the run that
      // follows needs more, its reserve moves the buffer and it falls back to its own k_init_guess launch
      SQD_TRY(c->X.reserve((size_t)c->dav_nvecs_hint * na * nb * 8));
      SQD_TRY(reserve_counters(c));
      job.x = c->X.as<double>();
      job.pmin = c->guess_min.as<double>();
      job.pidx = reinterpret_cast<const int64_t*>(job.pmin + nrows);
      job.nrows = (int)nrows;
      job.n = na * nb;
      job.st = static_cast<DavState*>(dav_state_ptr(c));
      job.counter = counter_ptr(c);
      c->guess_x = job.x;
    }
    FillArgs& fa = b.fill;
    fa.p = la;
    fa.h1 = c->h1.as<double>();
    fa.eri4 = c->eri4.as<double>();
    fa.norb = c->norb;
    fa.job = job;
    fa.gx = nblk(b.maxn, 4);
    b.have_fill = true;
  }
  return SQD_OK;
}

// phase 2, host planning: the sigma kernel of this subspace, the link arrays, the arguments of launch C (fill +
// decorate + the start of the Davidson run) and -- for the work-item kernel -- the descriptors and the work list.
// always_guess: prepare the Davidson run even when the subspace has no link at all (batched solves run launch C for
// every subspace; a single solve skips the launch then and the solver starts itself).
static int subspace_phase2_plan(sqd_ctx* c, SubspaceBuild& b, bool always_guess) {
  const int64_t na = b.na, nb = b.nb, row0 = b.row0, row1 = b.row1, nrows = b.nrows;
  const int64_t* tot = b.tot;
  const int* nocc = b.nocc;
  const int nnorb = c->nnorb;
  // Ultra-sparse coupling (at most two links per string on average, either spin): sigma is the element-gather
  // kernel.  SQD_SIGMA_DIRECT=1 / 0 forces / forbids it (tests run both kernels on the same inputs).
  // Long rows with short, even lists (uniform-random sets beyond ~10^3 strings per spin): k_sigma_rows, R whole rows
  // of C per workgroup in LDS.  It shares the element-gather kernel's table layout (CSR lists only) plus the beta
  // doubles in jagged-diagonal order.  SQD_SIGMA_ROWS=0 forbids it, =R forces R rows per workgroup.
  {
    bool direct = (tot[0] + tot[1] <= 2 * na) && (tot[2] + tot[3] <= 2 * nb);
    const char* env_d = std::getenv("SQD_SIGMA_DIRECT");
    if (env_d) direct = std::atoi(env_d) != 0;
    const size_t lds_rows = ((size_t)c->lds_bytes - 3 * 1024) / ((size_t)((nb + 1) & ~int64_t(1)) * 8);  // rows that fit
    int rows = 0;
    // (single links at most two per string on average: the single x single term is a nested per-element loop there)
    if (!direct && !env_d && lds_rows >= 1 && nb >= 1024 && tot[2] + tot[3] <= 32 * nb && tot[0] + tot[1] <= 64 * na &&
        tot[0] <= 2 * na && tot[2] <= 2 * nb)
      rows = (int)(lds_rows < 8 ? lds_rows : 8);
    if (const char* env = std::getenv("SQD_SIGMA_ROWS")) {
      rows = std::atoi(env);
      if (rows > 0 && lds_rows < 1) rows = 0;
      if (rows > (int)lds_rows) rows = (int)lds_rows;
      if (rows > 8) rows = 8;
    }
    // template instantiations: 1, 2, 3, 4, 6, 8
    if (rows == 5) rows = 4;
    if (rows == 7) rows = 6;
    c->sig_rows = rows > 0 ? rows : 0;
    c->sig_direct = direct || c->sig_rows > 0;
  }
  // launch C: fill + decorate (unless it went out already, behind launch A: build_subspace)
  for (int s = 0; s < 2; ++s) {
    c->sp[s].n_s = tot[2 * s];
    c->sp[s].n_d = tot[2 * s + 1];
  }
  if (!b.fill_launched) {
    int64_t maxl = 0;
    for (int k = 0; k < 4; ++k) maxl = tot[k] > maxl ? tot[k] : maxl;
    SQD_TRY(subspace_fill_setup(c, b, tot, maxl > 0 || always_guess));
  }
  c->na = na;
  c->nb = nb;
  c->row0 = row0;
  c->row1 = row1;
  c->D = na * nb;
  c->nelec[0] = nocc[0];
  c->nelec[1] = nocc[1];
  // large sets with short, even lists: the list passes (sqd_lists.hip) take the whole sigma; CSR-only table layout.
  // Single builds only: the batched solve has no launch class for it (such subspaces are solved one by one there).
  if (!always_guess && lists_select(c, na, nb, row0, row1, tot, nocc)) {
    c->sig_direct = true;
    c->sig_rows = 0;
  } else {
    c->sig_lists = false;
  }
  b.have_ell = b.have_jds = b.have_dense = false;
  b.ups.clear();
  b.blob_bytes = 0;
  // Dense same-spin blocks on the matrix cores: connected sets of batch size (the SQD loop's carry-over sets are 20-22 %
  // dense in both spins at 200-550 strings per spin; HF-centred synthetic sets 22-26 %).  Below ~8 % the sparse work
  // items do less work than the padded dense product; SQD_SIGMA_DENSE=1 / 0 forces / forbids (tests run both).
  {
    bool dense = !c->sig_direct && row0 == 0 && row1 == na && na <= 4096 && nb <= 4096 &&
                 (double)(tot[0] + tot[1]) >= 0.08 * (double)na * (double)na &&
                 (double)(tot[2] + tot[3]) >= 0.08 * (double)nb * (double)nb;
    if (const char* env = std::getenv("SQD_SIGMA_DENSE"))
      dense = std::atoi(env) != 0 && !c->sig_direct && row0 == 0 && row1 == na && na <= 4096 && nb <= 4096;
    // ... and from ~10^3 strings per spin, where the blocks have thinned out to 5-11 %, as a sparse product in
    // row-AXPY form (sqd_spmm.hip); single builds only (the batched solve has no launch class for it).  An explicit
    // SQD_SIGMA_DENSE wins over the default choice, SQD_SIGMA_SPMM=1 over both.
    const bool spmm = !always_guess && spmm_select(c, na, nb, row0, row1, tot, c->sig_direct) &&
                      (std::getenv("SQD_SIGMA_SPMM") || !std::getenv("SQD_SIGMA_DENSE"));
    if (!spmm) c->sig_spmm = false;
    if (spmm) dense = false;
    opp_select(c, na, nb, tot);  // (only with the sparse product: clears sig_opp otherwise)
    c->sig_dense = dense || spmm;
    if (dense) {
      c->dense_pa = (int)((na + 63) / 64 * 64);
      c->dense_pb = (int)((nb + 63) / 64 * 64);
      SQD_TRY(c->hdense_a.reserve((size_t)c->dense_pa * c->dense_pa * 8));
      SQD_TRY(c->hdense_b.reserve((size_t)c->dense_pb * c->dense_pb * 8));
      SQD_TRY(c->gdense.reserve((size_t)DENSE_SPLIT * na * nb * 8));  // the partial products
      DenseFillArgs& d = b.dense;
      for (int s = 0; s < 2; ++s) {
        const SpinTables& t = c->sp[s];
        d.n[s] = t.n;
        d.P[s] = s ? c->dense_pb : c->dense_pa;
        d.s_ptr[s] = t.s_ptr.as<int64_t>();
        d.d_ptr[s] = t.d_ptr.as<int64_t>();
        d.s_rec[s] = t.s_rec.as<SRec>();
        d.s_val[s] = t.s_val.as<double>();
        d.d_val[s] = t.d_val.as<double>();
        d.d_src[s] = t.d_src.as<uint32_t>();
        d.out[s] = (s ? c->hdense_b : c->hdense_a).as<double>();
      }
      d.gx = (unsigned)(c->dense_pa > c->dense_pb ? c->dense_pa : c->dense_pb);
      b.have_dense = true;
    }
  }
  if (c->sig_direct) {
    // the element-gather sigma kernel reads the CSR lists as they are: no work items, no ELL copies, no merged
    // same-spin list, no descriptor upload -- launch D does not exist
    c->n_items = c->n_multi = c->n_slots = 0;
    if (c->sig_rows) {
      SpinTables& t = c->sp[1];
      SQD_TRY(t.jd_src.reserve((size_t)t.n_d * 4));
      SQD_TRY(t.jd_val.reserve((size_t)t.n_d * 8));
      if (t.n_d > 0) {
        JdsArgs& j = b.jds;
        j.n = nb;
        j.d_ptr = t.d_ptr.as<int64_t>();
        j.d_src = t.d_src.as<uint32_t>();
        j.d_val = t.d_val.as<double>();
        j.jd_src = t.jd_src.as<uint32_t>();
        j.jd_val = t.jd_val.as<double>();
        j.gx = nblk((nb + 63) / 64, 4);
        b.have_jds = true;
      }
    }
    return SQD_OK;
  }
  {
    SpinTables& t = c->sp[0];  // merged same-spin CSR (singles then doubles of each row) for the row role's AXPY items
    SQD_TRY(t.hs_ptr.reserve((t.n + 1) * 8));
    SQD_TRY(t.hs_src.reserve((size_t)(t.n_s + t.n_d) * 4));
    SQD_TRY(t.hs_val.reserve((size_t)(t.n_s + t.n_d) * 8));
  }
  // capped sliced-ELL copies for the column role (beta): descriptors on the host, fill on the device
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
  const int64_t* dptr_b = c->h_dptr_b;
  if (c->sig_dense) {  // the beta doubles are in the dense block: empty lists for the work items
    b.zero_ptr.assign((size_t)nb + 1, 0);
    dptr_b = b.zero_ptr.data();
  }
  for (;;) {
    if (!lds_rows) chunk_cols = forced ? forced : 4096;
    const size_t target = lds_rows ? 40 * 1024 : 120 * 1024;
    for (cap = cap0;; cap *= 2) {
      make_vrows(c->h_sptr_b, nb, cap, chunk_cols, vs);
      make_vrows(dptr_b, nb, cap, chunk_cols, vd);
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
        make_vrows(dptr_b, nb, cap, chunk_cols, vd);
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
  if (c->sig_dense && vd.total != 0) {
    set_error("internal: dense same-spin mode with beta doubles left in the work items");
    return SQD_ERR_STATE;
  }
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
  SQD_TRY(build_sigma_work(c));
  // all descriptors and the work list travel in ONE host blob / ONE copy; the per-array DevBufs are views into it
  b.ups = {
      {&t.vs_cnt, vs.vcnt.data(), vs.vcnt.size() * 4},   {&t.vs_own, vs.own.data(), vs.own.size() * 4},
      {&t.vs_start, vs.vstart.data(), vs.vstart.size() * 8}, {&t.es_sl, vs.sl.data(), vs.sl.size() * 8},
      {&t.vd_cnt, vd.vcnt.data(), vd.vcnt.size() * 4},   {&t.vd_own, vd.own.data(), vd.own.size() * 4},
      {&t.vd_start, vd.vstart.data(), vd.vstart.size() * 8}, {&t.ed_sl, vd.sl.data(), vd.sl.size() * 8},
      {&t.vs_chunk, vs.chunk.data(), vs.chunk.size() * 4}, {&t.vd_chunk, vd.chunk.data(), vd.chunk.size() * 4},
      {&c->items, c->h_items.data(), c->h_items.size() * sizeof(WorkItem)},
      {&c->multi, c->h_multi.data(), c->h_multi.size() * sizeof(MultiRow)},
      {&c->rowinfo, c->h_rowinfo.data(), c->h_rowinfo.size() * 4},
  };
  size_t blob = 0;
  for (const auto& u : b.ups) blob += (u.bytes + 15) & ~size_t(15);
  b.blob_bytes = blob + 16;
  SQD_TRY(t.es_rec.reserve((size_t)vs.total * sizeof(SRec) + 8));
  SQD_TRY(t.es_val.reserve((size_t)vs.total * 8 + 8));
  SQD_TRY(t.ed_src.reserve((size_t)vd.total * 4 + 8));
  SQD_TRY(t.ed_val.reserve((size_t)vd.total * 8 + 8));
  b.have_ell = true;
  return SQD_OK;
}

// phase 2, placement: the descriptor blob into (h_blob, d_blob) -- pinned host memory and where it will be on the
// device -- and the arguments of launch D (merged alpha CSR + both capped-ELL copies)
static void subspace_phase2_place(sqd_ctx* c, SubspaceBuild& b, char* h_blob, char* d_blob) {
  if (!b.have_ell) return;
  size_t off = 0;
  for (const auto& u : b.ups) {
    if (u.bytes) std::memcpy(h_blob + off, u.src, u.bytes);
    u.buf->set_view(d_blob + off);
    off += (u.bytes + 15) & ~size_t(15);
  }
  const SpinTables& ta = c->sp[0];
  const SpinTables& t = c->sp[1];
  EllArgs& g = b.ell;
  g.n_a = ta.n;
  g.sa_ptr = ta.s_ptr.as<int64_t>();
  g.da_ptr = ta.d_ptr.as<int64_t>();
  g.sa_rec = ta.s_rec.as<SRec>();
  g.sa_val = ta.s_val.as<double>();
  g.da_src = ta.d_src.as<uint32_t>();
  g.da_val = ta.d_val.as<double>();
  g.hs_ptr = ta.hs_ptr.as<int64_t>();
  g.hs_src = ta.hs_src.as<uint32_t>();
  g.hs_val = ta.hs_val.as<double>();
  g.nv_s = t.n_s > 0 ? c->hv_s.nv : 0;
  g.nv_d = (t.n_d > 0 && !c->sig_dense) ? c->hv_d.nv : 0;
  g.vs_cnt = t.vs_cnt.as<int32_t>();
  g.vd_cnt = t.vd_cnt.as<int32_t>();
  g.vs_start = t.vs_start.as<int64_t>();
  g.vd_start = t.vd_start.as<int64_t>();
  g.es_sl = t.es_sl.as<int64_t>();
  g.ed_sl = t.ed_sl.as<int64_t>();
  g.sb_rec = t.s_rec.as<SRec>();
  g.sb_val = t.s_val.as<double>();
  g.db_src = t.d_src.as<uint32_t>();
  g.db_val = t.d_val.as<double>();
  g.es_rec = t.es_rec.as<SRec>();
  g.es_val = t.es_val.as<double>();
  g.ed_src = t.ed_src.as<uint32_t>();
  g.ed_val = t.ed_val.as<double>();
  const unsigned gx_m = nblk(ta.n + 1, 4), gx_s = nblk(g.nv_s, 256), gx_d = nblk(g.nv_d, 256);
  unsigned gx = gx_m > gx_s ? gx_m : gx_s;
  gx = gx > gx_d ? gx : gx_d;
  g.gx = gx;
}

int build_subspace(sqd_ctx* c, const uint64_t* sa, int64_t na, const uint64_t* sb, int64_t nb, int64_t row0,
                   int64_t row1) {
  hipStream_t st = c->stream;
  SQD_TRY(stage_reset(c));
  if (c->want_timing) SQD_HIP_CHECK(hipEventRecord(c->ev[0], st));
  SubspaceBuild b;
  SQD_TRY(subspace_phase1(c, b, sa, na, sb, nb, row0, row1, nullptr));
  {
    void* h = nullptr;  // both string lists: one staged upload
    SQD_TRY(stage_alloc(c, (size_t)(na + nb) * 8, &h));
    std::memcpy(h, sa, (size_t)na * 8);
    std::memcpy(static_cast<char*>(h) + (size_t)na * 8, sb, (size_t)nb * 8);
    SQD_HIP_CHECK(hipMemcpyAsync(c->strs2.p, h, (size_t)(na + nb) * 8, hipMemcpyHostToDevice, st));
  }
  hipLaunchKernelGGL(k_tables_count, dim3(b.count.gx, 4), dim3(256), 0, st, b.count);
  SQD_HIP_CHECK(hipGetLastError());
  // Single solves of batch size (both lists <= 1024 strings, all rows): the link arrays are reserved at their upper
  // bound -- every ordered pair of strings is at most one link: n (n - 1) records, 42 MB per spin at 1024 -- so launch C
  // needs nothing from the host and shares ONE launch with B, right behind A (k_tables_diag_fill)
  static const bool fused_env = [] {
    const char* env = std::getenv("SQD_TABLES_FUSED");
    return !env || std::atoi(env) != 0;
  }();
  const bool fused = fused_env && row0 == 0 && (row1 < 0 || row1 == na) && na <= 1024 && nb <= 1024 &&
                     (size_t)c->dav_nvecs_hint * na * nb * 8 <= (size_t(1) << 30);
  if (fused) {
    const int64_t cap[4] = {na * (na - 1) + 1, na * (na - 1) + 1, nb * (nb - 1) + 1, nb * (nb - 1) + 1};
    SQD_TRY(subspace_fill_setup(c, b, cap, true));
    DiagFillArgs df;
    df.d = b.diag;
    df.d.coherent_min = 1;
    df.f = b.fill;
    df.counter = counter3_ptr(c);
    const unsigned gx = df.d.gx > df.f.gx ? df.d.gx : df.f.gx;
    hipLaunchKernelGGL(k_tables_diag_fill, dim3(gx, 5), dim3(256), 0, st, df);
    SQD_HIP_CHECK(hipGetLastError());
    b.fill_launched = true;
  } else {
    // launch B is queued behind A and runs while the host waits for the pointers and cuts the work lists
    hipLaunchKernelGGL(k_tables_diag, dim3(b.diag.gx, 3), dim3(256), 0, st, b.diag);
    SQD_HIP_CHECK(hipGetLastError());
  }
  SQD_TRY(subspace_wait_pointers(c, b));
  SQD_TRY(subspace_phase2_plan(c, b, /*always_guess=*/false));
  if (b.have_fill && !b.fill_launched) {
    hipLaunchKernelGGL(k_tables_fill, dim3(b.fill.gx, 2), dim3(256), 0, st, b.fill);
    SQD_HIP_CHECK(hipGetLastError());
  }
  if (b.have_jds) {
    hipLaunchKernelGGL(k_tables_jds, dim3(b.jds.gx), dim3(256), 0, st, b.jds);
    SQD_HIP_CHECK(hipGetLastError());
  }
  if (c->sig_lists) SQD_TRY(lists_build(c));
  if (c->sig_spmm) SQD_TRY(spmm_build(c));
  if (c->sig_opp) SQD_TRY(opp_build(c));
  if (b.have_dense) {
    hipLaunchKernelGGL(k_tables_dense, dim3(b.dense.gx, 2), dim3(256), 0, st, b.dense);
    SQD_HIP_CHECK(hipGetLastError());
  }
  if (b.have_ell) {
    void* h_blob = nullptr;
    SQD_TRY(stage_alloc(c, b.blob_bytes, &h_blob));
    SQD_TRY(c->d_blob.reserve(b.blob_bytes));
    subspace_phase2_place(c, b, static_cast<char*>(h_blob), static_cast<char*>(c->d_blob.p));
    if (b.blob_bytes > 16)
      SQD_HIP_CHECK(hipMemcpyAsync(c->d_blob.p, h_blob, b.blob_bytes - 16, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_tables_ell, dim3(b.ell.gx, 3), dim3(256), 0, st, b.ell);
    SQD_HIP_CHECK(hipGetLastError());
  }
  // no synchronisation here: later calls use the same stream; ev[0]..ev[1] is read lazily
  if (c->want_timing) SQD_HIP_CHECK(hipEventRecord(c->ev[1], st));
  c->stage_pending = true;
  c->ms_setup = c->want_timing ? -1.0 : 0.0;
  c->have_subspace = true;
  return SQD_OK;
}

// ---- batched table build (sqd_solve_batch): the phases of all subspaces in lock-step; the kernels of one dependency
// level of ALL subspaces go out as one launch (blockIdx.z = subspace), their arguments, the strings and the
// descriptor blobs travel in ONE host-to-device copy per phase.  Per-subspace results are the single build's, bit
// for bit: the same bodies on the same per-subspace grids.
int BatchStage::reserve(size_t bytes) {
  if (bytes <= cap) return SQD_OK;
  if (host) SQD_HIP_CHECK(hipHostFree(host));
  host = nullptr;
  cap = 0;
  const size_t want = bytes + bytes / 2 + 4096;
  SQD_HIP_CHECK(hipHostMalloc((void**)&host, want, hipHostMallocDefault));
  SQD_TRY(dev.reserve(want));
  cap = want;
  return SQD_OK;
}
void BatchStage::release() {
  if (host) {
    hipError_t e = hipHostFree(host);
    (void)e;
  }
  host = nullptr;
  cap = 0;
  dev.release();
}

template <class T>
static size_t stage_take(size_t& off, size_t count) {
  off = (off + 63) & ~size_t(63);
  const size_t at = off;
  off += count * sizeof(T);
  return at;
}

int build_subspace_batch(sqd_ctx* parent, const std::vector<sqd_ctx*>& subs, const uint64_t* const* sa,
                         const int64_t* na, const uint64_t* const* sb, const int64_t* nb) {
  const int n = (int)subs.size();
  hipStream_t st = parent->stream;
  std::vector<SubspaceBuild> bs(n);
  // ---- phase 1: [CountArgs[n] | DiagArgs[n] | strings of every subspace]
  BatchStage& s1 = parent->bstage[0];
  size_t off = 0;
  const size_t o_count = stage_take<CountArgs>(off, n), o_diag = stage_take<DiagArgs>(off, n);
  std::vector<size_t> o_str(n);
  for (int p = 0; p < n; ++p) {
    if (na[p] <= 0 || nb[p] <= 0 || !sa[p] || !sb[p]) {
      set_error("empty CI string list in batch " + std::to_string(p));
      return SQD_ERR_INVALID;
    }
    o_str[p] = stage_take<uint64_t>(off, (size_t)(na[p] + nb[p]));
  }
  SQD_TRY(s1.reserve(off));
  char* d1 = static_cast<char*>(s1.dev.p);
  unsigned gx_count = 0, gx_diag = 0;
  for (int p = 0; p < n; ++p) {
    sqd_ctx* c = subs[p];
    c->want_timing = false;
    SQD_TRY(subspace_phase1(c, bs[p], sa[p], na[p], sb[p], nb[p], 0, -1, reinterpret_cast<uint64_t*>(d1 + o_str[p])));
    std::memcpy(s1.host + o_str[p], sa[p], (size_t)na[p] * 8);
    std::memcpy(s1.host + o_str[p] + (size_t)na[p] * 8, sb[p], (size_t)nb[p] * 8);
    reinterpret_cast<CountArgs*>(s1.host + o_count)[p] = bs[p].count;
    reinterpret_cast<DiagArgs*>(s1.host + o_diag)[p] = bs[p].diag;
    gx_count = bs[p].count.gx > gx_count ? bs[p].count.gx : gx_count;
    gx_diag = bs[p].diag.gx > gx_diag ? bs[p].diag.gx : gx_diag;
  }
  SQD_HIP_CHECK(hipMemcpyAsync(d1, s1.host, off, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_tables_count_b, dim3(gx_count, 4, n), dim3(256), 0, st,
                     reinterpret_cast<const CountArgs*>(d1 + o_count));
  SQD_HIP_CHECK(hipGetLastError());
  hipLaunchKernelGGL(k_tables_diag_b, dim3(gx_diag, 3, n), dim3(256), 0, st,
                     reinterpret_cast<const DiagArgs*>(d1 + o_diag));
  SQD_HIP_CHECK(hipGetLastError());
  // ---- phase 2: [FillArgs[n] | EllArgs[n_ell] | JdsArgs[n_jds] | descriptor blobs]
  for (int p = 0; p < n; ++p) {
    SQD_TRY(subspace_wait_pointers(subs[p], bs[p]));
    SQD_TRY(subspace_phase2_plan(subs[p], bs[p], /*always_guess=*/true));
  }
  BatchStage& s2 = parent->bstage[1];
  off = 0;
  int n_ell = 0, n_jds = 0, n_dense = 0;
  for (int p = 0; p < n; ++p) {
    n_ell += bs[p].have_ell;
    n_jds += bs[p].have_jds;
    n_dense += bs[p].have_dense;
  }
  const size_t o_fill = stage_take<FillArgs>(off, n), o_ell = stage_take<EllArgs>(off, n_ell),
               o_jds = stage_take<JdsArgs>(off, n_jds), o_dense = stage_take<DenseFillArgs>(off, n_dense);
  std::vector<size_t> o_blob(n, 0);
  for (int p = 0; p < n; ++p)
    if (bs[p].have_ell) o_blob[p] = stage_take<char>(off, bs[p].blob_bytes);
  SQD_TRY(s2.reserve(off));
  char* d2 = static_cast<char*>(s2.dev.p);
  unsigned gx_fill = 0, gx_ell = 0, gx_jds = 0, gx_dense = 0;
  int i_ell = 0, i_jds = 0, i_dense = 0;
  for (int p = 0; p < n; ++p) {
    subspace_phase2_place(subs[p], bs[p], s2.host + o_blob[p], d2 + o_blob[p]);
    reinterpret_cast<FillArgs*>(s2.host + o_fill)[p] = bs[p].fill;
    gx_fill = bs[p].fill.gx > gx_fill ? bs[p].fill.gx : gx_fill;
    if (bs[p].have_ell) {
      reinterpret_cast<EllArgs*>(s2.host + o_ell)[i_ell++] = bs[p].ell;
      gx_ell = bs[p].ell.gx > gx_ell ? bs[p].ell.gx : gx_ell;
    }
    if (bs[p].have_jds) {
      reinterpret_cast<JdsArgs*>(s2.host + o_jds)[i_jds++] = bs[p].jds;
      gx_jds = bs[p].jds.gx > gx_jds ? bs[p].jds.gx : gx_jds;
    }
    if (bs[p].have_dense) {
      reinterpret_cast<DenseFillArgs*>(s2.host + o_dense)[i_dense++] = bs[p].dense;
      gx_dense = bs[p].dense.gx > gx_dense ? bs[p].dense.gx : gx_dense;
    }
  }
  SQD_HIP_CHECK(hipMemcpyAsync(d2, s2.host, off, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(k_tables_fill_b, dim3(gx_fill, 2, n), dim3(256), 0, st,
                     reinterpret_cast<const FillArgs*>(d2 + o_fill));
  SQD_HIP_CHECK(hipGetLastError());
  if (n_jds) {
    hipLaunchKernelGGL(k_tables_jds_b, dim3(gx_jds, 1, n_jds), dim3(256), 0, st,
                       reinterpret_cast<const JdsArgs*>(d2 + o_jds));
    SQD_HIP_CHECK(hipGetLastError());
  }
  if (n_dense) {
    hipLaunchKernelGGL(k_tables_dense_b, dim3(gx_dense, 2, n_dense), dim3(256), 0, st,
                       reinterpret_cast<const DenseFillArgs*>(d2 + o_dense));
    SQD_HIP_CHECK(hipGetLastError());
  }
  if (n_ell) {
    hipLaunchKernelGGL(k_tables_ell_b, dim3(gx_ell, 3, n_ell), dim3(256), 0, st,
                       reinterpret_cast<const EllArgs*>(d2 + o_ell));
    SQD_HIP_CHECK(hipGetLastError());
  }
  for (int p = 0; p < n; ++p) {
    subs[p]->ms_setup = 0.0;
    subs[p]->have_subspace = true;
  }
  return SQD_OK;
}

}  // namespace sqd
