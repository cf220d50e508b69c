// Opposite-spin part of sigma (and the diagonal) for CONNECTED string sets of 10^3 strings per spin and more, in front of
// which sqd_spmm.hip has formed the same-spin product G = H_a C + C H_b:
//   sigma[A,B] = hdiag[A,B] C[A,B] + G[A,B]
//     + sum_{(A',pq,s) in Sa(A)} s * Jb[B][pq] * C[A',B]            alpha single x beta occupation
//     + sum_{(B',rs,t) in Sb(B)} t * Ja[A][rs] * C[A,B']            beta single x alpha occupation
//     + sum_{Sa(A)} sum_{Sb(B)} s t (pq|rs) C[A',B']                single x single
// -- what pyscf's selected_ci.contract_2e evaluates through SCIcontract_2e_bbaa (reference call sites
// qiskit_addon_sqd/fermion.py:721-723, :810-818).  The last two terms are one sum over the "entries" of
// row A -- the row itself (weights Ja[A][:]) and its alpha single links (weights (pq|:)) -- times the beta single links.
//
// Round 6: the formulation for rows of more than 3072 columns (k_opp_src).  k_opp_rows (sqd_opp.hip) keeps the beta links
// by TARGET column range, every range staging whole source rows: its long-row instantiations (4-8 staged columns per
// thread) spilled 6-40 registers and re-staged every source row once per range (4.7 GB per sigma at 3000 x 3000 already).
// Measured against it (profiles/r06/opp_src_probe.txt): 5 % slower at 5000 x 5000, 23 % at 3000 x 3000, 47 % at 1000 x
// 1000 -- the short rows stay with k_opp_rows, whose <= 3-column instantiations do not spill.  ONE workgroup owns a piece of a target row A (<= E = 16 of its entries) and walks the beta
// link list in PASSES over ranges of the SOURCE column B':
//   * a pass stages its range [q0, q1) of ALL the piece's source rows at once -- one round trip for the whole pass, every
//     element of a source row staged once per item, not once per range -- as E / 2 planes of interleaved pairs
//     Cst[plane][B' - q0][2] (signed); the alpha single x beta occupation term rides on the staging pass;
//   * inside a range the links are grouped by excitation operator (widx = orbital pair and direction) in sub-runs of four:
//     a thread holds <= NSUB sub-runs -- per link ONE register, the byte offset of its source column in a plane, and one
//     accumulator -- and per sub-run its widx: a link costs one 16-byte LDS gather and two multiply-adds per plane, with
//     the plane as an immediate of the instruction; the weights (pq|rs) / Ja[A][rs] come per sub-run and plane straight
//     from the integral table (one plane ahead; the table is L2-resident), and the linear spin penalty is one addition to
//     the weight of the sub-run whose widx is the alpha link's partner;
//   * no barrier inside the plane loop: three per pass (staged / gathered / folded);
//   * at the end of a pass the per-link sums go through LDS to the threads that own the target columns (positions in
//     target order precomputed per range; runs summed in that order), and a thread carries its columns' sums over the
//     passes in registers: one sigma row per item, written once, same bits on every run.
// Rows in one piece are written in place; the others as partial rows that the first reader of the vector adds in slot
// order (k_dots_s inside a Davidson run, k_opp_src_reduce otherwise).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "sqd_common.h"

namespace sqd {

constexpr int OPPS_SUB = 4;       // links per sub-run (one weight pair serves four links)
constexpr int OPPS_NSUB_MAX = 2;  // sub-runs per thread (8 links: 8 + 16 registers; three sub-runs spill at 128 registers)
constexpr int OPPS_RMAX = 8;      // target columns per thread (nb <= OPPS_RMAX * threads)
constexpr int OPPS_RL = 6;        // ... of which, beside two sub-runs, at most this many are carried in LDS (2 of 5, 4 of 6, 6 of 8)
// LDS plan (bytes; the plane of a batch is an immediate of the gather instruction):
//   Cst[4][COLS][2] -- the staged range of eight entries (a piece is walked eight entries at a time); accb[links of a pass]
//   takes its place at the end of a pass; Wst[nnorb][8] -- the eight entries' weight rows -- and jbuf[T] sit behind.
//   BIG = false: planes of 512 columns, 32 + 32 + 4 KB (512 threads, two workgroups per CU, norb <= 31); BIG = true:
//   planes of 1024 columns, 64 + 64 + 8 KB (1024 threads: rows of more than 4096 columns; or norb <= 44)
template <bool BIG>
struct OppSrcLds {
  static constexpr int COLS = BIG ? 1024 : 512, NB = 4;
  static constexpr int PLANE = COLS * 16;
  static constexpr int STAGE_BYTES = NB * PLANE;
  static constexpr int WROWS = BIG ? 1024 : 512;     // orbital pairs (nnorb): norb <= 44 / 31
  static constexpr int WST = STAGE_BYTES;            // Wst[pair][8 entries]: the round's weights, 64 bytes per pair
};
constexpr int OPPS_EMAX = 64;  // entries of a piece: one per lane of a wavefront
// a link's table word: byte offset of its source column in a plane (bits 0-13) | its position among the pass's links in
// target order (bits 14-26) | sign of the beta link (27) | live (28); 0 = padding of a sub-run
constexpr uint32_t OPPS_ADDR = 0x3fffu, OPPS_POS_SHIFT = 14, OPPS_POS = 0x1fffu, OPPS_NEG = 1u << 27, OPPS_LIVE = 1u << 28;

// one workgroup's share of a row: entries [e0, e0 + ne) of row A (entry 0 = the row itself, entry e > 0 = its alpha single
// link e - 1); slot < 0: the row has this one item and is written in place, else partial row `slot` (added in slot
// order by the first reader of the vector -- k_dots_eig inside a Davidson run, k_opp_src_reduce otherwise)
struct OppSrcItem {
  uint32_t A;
  int32_t e0, ne, slot;
};
struct OppSrcState {
  DevBuf tab, colcut, items, rowinfo, partial, multi;
  std::vector<OppSrcItem> h_items;
  std::vector<int32_t> h_rowinfo;
  std::vector<MultiRow> h_multi;
  std::vector<uint32_t> h_tab;
  std::vector<int32_t> h_colcut;
  std::vector<SRec> h_rec;
  std::vector<uint32_t> h_row;
  int H = 1, nsub = 2, T = 512;
  bool big = false;
  int64_t n_items = 0, n_slots = 0, n_multi = 0;
  size_t shmem = 0;
};

void oppsrc_release(sqd_ctx* c) {
  if (!c->oppsrc) return;
  OppSrcState* s = static_cast<OppSrcState*>(c->oppsrc);
  for (DevBuf* b : {&s->tab, &s->colcut, &s->items, &s->rowinfo, &s->partial, &s->multi}) b->release();
  delete s;
  c->oppsrc = nullptr;
}

struct OppSrcArgs {
  GPtr<const double> c;
  GPtr<double> sigma, partial;
  GPtr<const double> hdiag, gdense, ja_row, jbT, eri_pp;
  GPtr<const int64_t> sa_ptr;
  GPtr<const SRec> sa_rec;
  GPtr<const uint32_t> tab;    // per pass: rec[S][T] | widx[NSUB][T] | colq[OPPS_RMAX][T]; colq = first position (target
                               // order) of column t + r T's links inside the pass | their number << 16
  GPtr<const int32_t> colcut;  // [H + 1] first source column of every pass
  GPtr<const OppSrcItem> items;
  int64_t nb;
  int nnorb, T, H, jbuf_off;
  unsigned n_items;
  GPtr<const int> stop, vec_index;
  int64_t c_stride, s_stride;
  // the linear spin penalty, sigma = (H + shift (S^2 - ss)) c (pyscf's fix_spin_ form for ss < sz(sz+1) + 0.1):
  // S^2 = sz(sz+1) + sum_p n_pb (1 - n_pa) - sum_{p != q} Ea_qp Eb_pq -- a diagonal term on the own row and -shift on the
  // weight of the beta links with the alpha link's orbital pair and the opposite direction
  int spin;
  double ss, shift, szterm;
  GPtr<const uint64_t> strs_a, strs_b;
};

// lane e of every wavefront holds entry e of the piece; a field of entry e reaches the scalar registers through
// v_readlane with e a constant of the unrolled loops -- no table in memory, no load behind a barrier
__device__ inline uint32_t oppsrc_lane(uint32_t v, int e) { return (uint32_t)__builtin_amdgcn_readlane((int)v, e); }
// element at a 32-bit BYTE offset of a row whose address is wave-uniform: scalar base + 32-bit vector offset -- one
// offset register for all the loads of a round instead of a 64-bit address pair per load
// ... the uniform base pinned to scalar registers behind an opaque lane read: left visible, the compiler folds the
// uniform row offset into the vector index and builds a 64-bit address pair per load after all
__device__ inline const double* oppsrc_pin(const double* p) {
  const uint64_t a = (uint64_t)(uintptr_t)p;
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  return reinterpret_cast<const double*>((uintptr_t)(((uint64_t)hi << 32) | lo));
}
__device__ inline double oppsrc_ldu(const double* base, uint32_t byte_off) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + byte_off);
}
// ... and is read again wherever it is used: left alone the compiler hoists all 6 x 16 lane reads out of the pass loop and
// keeps them in scalar registers it does not have (106 SGPRs, the rest spilled into vector lanes)
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
#define OPPS_REREAD(v) asm volatile("" : "+v"(v))
#else
#define OPPS_REREAD(v) ((void)(v))
#endif

// ---- phase clocks (probe builds only: -DSQD_PHASE_CLOCK; profiles/probes/_oppsrc_clock.py): thread 0 of every workgroup adds
// the 100 MHz wall-clock deltas of its phases into ITS OWN row of a device array; the host sums the rows
#ifdef SQD_PHASE_CLOCK
constexpr int OCLK_ROWS = 65536, OCLK_COLS = 8;
__device__ unsigned long long sqd_clk_oppsrc[OCLK_ROWS * OCLK_COLS];
#define OCLK(var) const unsigned long long var = wall_clock64()
#define OCLK_ADD(slot, d) do { if (threadIdx.x == 0) sqd_clk_oppsrc[(blockIdx.x % OCLK_ROWS) * OCLK_COLS + (slot)] += (unsigned long long)(d); } while (0)
#else
#define OCLK(var)
#define OCLK_ADD(slot, d)
#endif

template <int NSUB, int RM, bool BIG>
__global__ void __launch_bounds__(1024) k_opp_src(const OppSrcArgs g) {
  constexpr int S = OPPS_SUB * NSUB;
  using Lds = OppSrcLds<BIG>;
  constexpr int NB = Lds::NB, E = 2 * NB;  // planes, entries per round of a pass
  HIP_DYNAMIC_SHARED(double, smem)
  char* const lds = reinterpret_cast<char*>(smem);
  if (g.stop && *g.stop) return;
  const unsigned item_index = blockIdx.x;
  if (item_index >= g.n_items) return;
  OCLK(k_item0);
  const int T = g.T, tid = threadIdx.x, lane = tid & 63;
  OppSrcItem it = g.items[item_index];
  it.A = (uint32_t)__builtin_amdgcn_readfirstlane((int)it.A);  // (uniform: scalar registers, not a vector register each)
  it.e0 = __builtin_amdgcn_readfirstlane(it.e0);
  it.ne = __builtin_amdgcn_readfirstlane(it.ne);
  it.slot = __builtin_amdgcn_readfirstlane(it.slot);
  const int64_t A = it.A;
  const int64_t nb = g.nb;
  const int nn = g.nnorb;
  const int64_t vsel = g.vec_index ? (int64_t)__builtin_amdgcn_readfirstlane(*g.vec_index - 1) : 0;
  const double* __restrict__ C = g.c + vsel * g.c_stride;
  double* __restrict__ sig = g.sigma + vsel * g.s_stride;
  const int ne = it.ne;
  const bool spin = g.spin != 0;
  const double pen = -g.shift;
  // ---- the piece's entries, one per lane: source row, weight row (as an element offset from eri_pp: the J row of the
  // own-row entry lies in another array, its distance to eri_pp is added), orbital pair, sign, S^2 partner
  uint32_t en_src = (uint32_t)A, en_pair = 0u, en_wlo = 0u, en_whi = 0u, en_flags = 0u;  // flags: 1 valid, 2 link, 4 negative
  uint32_t en_part = 0xffffffffu;
  {
    const int e = it.e0 + lane;
    const bool valid = lane < ne, lnk = valid && e > 0;
    SRec r = SRec{(uint32_t)A, 0u};
    if (lnk) r = g.sa_rec[g.sa_ptr[A] + e - 1];
    const uint32_t widx = srec_widx(r.meta);
    en_src = r.src;
    en_pair = widx >> 1;
    const double* wrow = lnk ? (const double*)g.eri_pp + (int64_t)en_pair * nn : (const double*)g.ja_row + A * nn;
    const uint64_t wa = (uint64_t)(uintptr_t)wrow;
    en_wlo = (uint32_t)wa;
    en_whi = (uint32_t)(wa >> 32);
    en_flags = (valid ? 1u : 0u) | (lnk ? 2u : 0u) | ((lnk && (r.meta >> 31)) ? 4u : 0u);
    if (lnk) en_part = widx ^ 1u;  // same orbital pair, opposite direction
  }
  // a thread's target columns' sums, carried over the passes: in registers, but for the last RL of six beside two
  // sub-runs (the allocator spilled them anyway: 28-44 bytes of scratch) -- those sit in LDS behind jbuf
  constexpr int RL = (NSUB == 2 && RM >= 5) ? (RM == 5 ? 2 : RM - 2) : 0, RR = RM - RL;  // (2 of 5, 4 of 6, 6 of 8)
  static_assert(RL <= OPPS_RL, "cbuf holds OPPS_RL columns per thread");
  double colacc[RR > 0 ? RR : 1];
#pragma unroll
  for (int r = 0; r < RR; ++r) colacc[r] = 0.0;
  double* const accb = smem;
  double* const jbuf = reinterpret_cast<double*>(lds + g.jbuf_off);  // behind the nnorb pairs of Wst that are in use
  double* const cbuf = jbuf + T;  // [RL][T]
#pragma unroll
  for (int r = 0; r < RL; ++r) cbuf[r * T + tid] = 0.0;

  // a pass's tables: requested during the fold of the pass before (the first: here), so that no pass starts -- and no fold
  // runs -- behind a table round trip
  // (six columns beside two sub-runs: the column table would be six registers too many through the gather -- requested
  // with the scatter instead)
  constexpr bool CQ_EARLY = !(NSUB == 2 && RM >= 5);
  uint32_t rec[S], wq[NSUB], colq[RM];  // wq: byte offset of the sub-run's orbital pair in Wst | direction bit
  auto load_tables = [&](int h) {
    const uint32_t* __restrict__ tab = g.tab + (int64_t)h * ((S + NSUB + OPPS_RMAX) * (int64_t)T) + tid;
#pragma unroll
    for (int s = 0; s < S; ++s) rec[s] = tab[(int64_t)s * T];
#pragma unroll
    for (int j = 0; j < NSUB; ++j) {
      const uint32_t widx = tab[(int64_t)(S + j) * T];
      wq[j] = (widx >> 1) * 64u | (widx & 1u);
    }
    if constexpr (CQ_EARLY) {
#pragma unroll
      for (int r = 0; r < RM; ++r) colq[r] = tab[(int64_t)(S + NSUB + r) * T];
    }
  };
  if (CQ_EARLY) load_tables(0);
  for (int h = 0; h < g.H; ++h) {
    const int q0 = g.colcut[h], q1 = g.colcut[h + 1];
    if (!CQ_EARLY) load_tables(h);
    double acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = 0.0;
    double jacc = 0.0;
    const bool mine = q0 + tid < q1;
    const uint32_t Bc8 = (uint32_t)(mine ? q0 + tid : q1 - 1) * 8u;  // byte offset of the staged column in a row
    for (int eb = 0; eb < ne; eb += E) {  // eight entries at a time
      if (eb > 0) __syncthreads();  // (the planes are gathered: they may be overwritten)
      OCLK(k_r0);
      // ---- stage the range of eight source rows (signed, pairs interleaved); the alpha single x beta occupation term of
      // the staged column on the way.  Unconditional loads (a dead entry reads row A / J row 0): all 16 in flight together.
      {
        double x[E], jv[E];
#pragma unroll
        for (int i = 0; i < E; ++i) {
          const uint32_t src = oppsrc_lane(en_src, eb + i), pair = oppsrc_lane(en_pair, eb + i);
          const double* __restrict__ srow = oppsrc_pin(C + (int64_t)src * nb);
          const double* __restrict__ jrow = oppsrc_pin((const double*)g.jbT + (int64_t)pair * nb);
          x[i] = oppsrc_ldu(srow, Bc8);
          jv[i] = oppsrc_ldu(jrow, Bc8);
        }
#pragma unroll
        for (int i = 0; i < E; i += 2) {
          const uint32_t f0 = oppsrc_lane(en_flags, eb + i), f1 = oppsrc_lane(en_flags, eb + i + 1);
          const double x0 = (f0 & 1u) ? ((f0 & 4u) ? -x[i] : x[i]) : 0.0, x1 = (f1 & 1u) ? ((f1 & 4u) ? -x[i + 1] : x[i + 1]) : 0.0;
          jacc += (f0 & 2u) ? jv[i] * x0 : 0.0;
          jacc += (f1 & 2u) ? jv[i + 1] * x1 : 0.0;
          if (mine) *reinterpret_cast<double2*>(lds + (i >> 1) * Lds::PLANE + tid * 16) = make_double2(x0, x1);
        }
      }
      // ... and the eight entries' weight rows, Wst[pair][entry]: (pq_e | pair) for a link entry, Ja[A][pair] for the row
      // itself -- straight from the integral table (L2-resident), requested with the rows above
      const double* wrow[E];  // (read from the lanes by every thread: wave-uniform, and the emulator's lane exchange
                              //  needs the whole wavefront -- not inside the loop over pairs, which tid >= nnorb skip)
#pragma unroll
      for (int i = 0; i < E; ++i)
        wrow[i] = reinterpret_cast<const double*>((uintptr_t)(((uint64_t)oppsrc_lane(en_whi, eb + i) << 32) | oppsrc_lane(en_wlo, eb + i)));
      for (int p = tid; p < nn; p += T) {
        double wv[E];
#pragma unroll
        for (int i = 0; i < E; ++i) wv[i] = oppsrc_ldu(wrow[i], (uint32_t)p * 8u);  // (a dead entry: the lane's default row, times a zero plane)
#pragma unroll
        for (int i = 0; i < E; i += 2) *reinterpret_cast<double2*>(lds + Lds::WST + p * 64 + i * 8) = make_double2(wv[i], wv[i + 1]);
      }
      __syncthreads();
      OCLK(k_r1);
      OCLK_ADD(1, k_r1 - k_r0);
      // ---- gather: plane b = entries eb + 2b, eb + 2b + 1; per sub-run one more 16-byte read, its weight pair
      const int nbat = (ne - eb + 1) >> 1;  // planes that hold an entry
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (b < nbat) {  // (uniform)
          uint32_t p0 = 0xffffffffu, p1 = 0xffffffffu;
          if (spin) {  // (the partner widx in wq's encoding; 0xffffffff -- no partner -- stays out of reach)
            p0 = oppsrc_lane(en_part, eb + 2 * b);
            p1 = oppsrc_lane(en_part, eb + 2 * b + 1);
            p0 = p0 == 0xffffffffu ? p0 : ((p0 >> 1) * 64u | (p0 & 1u));
            p1 = p1 == 0xffffffffu ? p1 : ((p1 >> 1) * 64u | (p1 & 1u));
          }
#pragma unroll
          for (int j = 0; j < NSUB; ++j) {
            double2 w2 = *reinterpret_cast<const double2*>(lds + Lds::WST + b * 16 + (wq[j] & ~1u));
            if (spin) {
              w2.x += (wq[j] == p0) ? pen : 0.0;
              w2.y += (wq[j] == p1) ? pen : 0.0;
            }
#pragma unroll
            for (int k = 0; k < OPPS_SUB; ++k) {
              const double2 c2 = *reinterpret_cast<const double2*>(lds + b * Lds::PLANE + (rec[OPPS_SUB * j + k] & OPPS_ADDR));
              acc[OPPS_SUB * j + k] += w2.x * c2.x;
              acc[OPPS_SUB * j + k] += w2.y * c2.y;
            }
          }
        }
      }
#ifdef SQD_PHASE_CLOCK
      { OCLK(k_r2); __builtin_amdgcn_s_waitcnt(0); OCLK(k_r3); OCLK_ADD(2, k_r3 - k_r1); OCLK_ADD(7, 1); (void)k_r2; }
#endif
    }
    OCLK(k_f0);
    __syncthreads();  // every gather of the pass is done: the planes become accb
    // ---- per-link sums -> target columns: every live link's sum to its position in target order (signed); the owner of
    // a column adds its run in that order, then the staged column's J term.
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const uint32_t p = rec[s];
      if (p & OPPS_LIVE) accb[(p >> OPPS_POS_SHIFT) & OPPS_POS] = (p & OPPS_NEG) ? -acc[s] : acc[s];
    }
    uint32_t cq[CQ_EARLY ? RM : 1];
    if constexpr (CQ_EARLY) {
#pragma unroll
      for (int r = 0; r < RM; ++r) cq[r] = colq[r];
    }
    const uint32_t* __restrict__ cqtab = g.tab + (int64_t)h * ((S + NSUB + OPPS_RMAX) * (int64_t)T) + (int64_t)(S + NSUB) * T + tid;
    if (CQ_EARLY && h + 1 < g.H) load_tables(h + 1);  // (lands while the columns are summed)
    jbuf[tid] = jacc;
    __syncthreads();
    OCLK(k_f1);
    OCLK_ADD(3, k_f1 - k_f0);
    auto column_sum = [&](int r) -> double {  // the run of column tid + r T in accb, in order, + the staged column's J term
      const int64_t B = tid + (int64_t)r * T;
      const uint32_t cqv = (CQ_EARLY && r < RR) ? cq[(CQ_EARLY && r < RR) ? r : 0] : cqtab[(int64_t)r * T];
      const uint32_t c0 = cqv & 0xffffu, n = cqv >> 16;
      double sum = 0.0;
      for (uint32_t i0 = 0; i0 < n; i0 += 4) {  // four reads in flight, added in order
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = (i0 + u < n) ? accb[c0 + i0 + u] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) sum += v[u];
      }
      if (B >= q0 && B < q1) sum += jbuf[B - q0];
      return sum;
    };
#pragma unroll
    for (int r = 0; r < RR; ++r)  // the columns carried in registers
      if (tid + (int64_t)r * T < nb) colacc[r] += column_sum(r);
    if constexpr (RL > 0) {       // ... and in LDS (eight columns: two at a time -- fully unrolled they spilled)
      constexpr int UNR = 2;
#pragma unroll UNR
      for (int r = RR; r < RM; ++r)
        if (tid + (int64_t)r * T < nb) cbuf[(r - RR) * T + tid] += column_sum(r);
    }
    __syncthreads();  // (accb / jbuf are read: the next pass may stage)
    OCLK(k_f2);
    OCLK_ADD(4, k_f2 - k_f1);
    OCLK_ADD(0, 1);
  }
  const bool has0 = it.e0 == 0;  // the piece that holds the row itself also brings the diagonal and the same-spin product
  const double* __restrict__ crow = C + A * nb;
  const double* __restrict__ hd = g.hdiag + A * nb;
  const double* __restrict__ gd = g.gdense + A * nb;
  double* __restrict__ orow = it.slot < 0 ? sig + A * nb : g.partial + (int64_t)it.slot * nb;
  auto finish = [&](int r, double v) {
    const int64_t B = tid + (int64_t)r * T;
    if (B < nb) {
      if (has0) {
        double d = hd[B];
        if (spin) d += g.shift * (g.szterm + (double)__popcll(g.strs_b[B] & ~g.strs_a[A]) - g.ss);
        v += d * crow[B] + gd[B];
      }
      orow[B] = v;
    }
  };
#pragma unroll
  for (int r = 0; r < RR; ++r) finish(r, colacc[r]);
  if constexpr (RL > 0) {
#pragma unroll 1
    for (int r = RR; r < RM; ++r) finish(r, cbuf[(r - RR) * T + tid]);
  }
  OCLK(k_item1);
  OCLK_ADD(5, k_item1 - k_item0);
  OCLK_ADD(6, 1);
}

// sigma[A, :] = sum of the partial rows of A in slot order, for the rows that were cut into several items (outside
// Davidson runs; inside, k_dots_eig adds them as the first reader of the vector)
struct OppSrcReduceArgs {
  GPtr<const MultiRow> rows;
  GPtr<const double> partial;
  GPtr<double> sigma;
  int64_t nb;
  GPtr<const int> stop, vec_index;
  int64_t s_stride;
};
__global__ void __launch_bounds__(256) k_opp_src_reduce(const OppSrcReduceArgs g) {
  if (g.stop && *g.stop) return;
  const MultiRow mr = g.rows[blockIdx.x];
  double* __restrict__ sig = g.sigma + (g.vec_index ? (int64_t)(*g.vec_index - 1) * g.s_stride : 0) + (int64_t)mr.A * g.nb;
  for (int64_t B = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; B < g.nb; B += (int64_t)gridDim.y * blockDim.x) {
    double sacc = 0.0;
    for (int j = 0; j < mr.nslots; ++j) sacc += g.partial[(int64_t)(mr.slot0 + j) * g.nb + B];
    sig[B] = sacc;
  }
}

// ---- host side
static int oppsrc_jbuf(bool big, int nnorb) { return (big ? OppSrcLds<true>::WST : OppSrcLds<false>::WST) + ((nnorb + 1) & ~1) * 64; }
static size_t oppsrc_shmem(bool big, int nnorb, int T, int nsub, int64_t nb) {  // planes (accb in their place at the end of a pass) | Wst | jbuf[T] | cbuf[RL][T]
  const int rm = (int)((nb + T - 1) / T);
  const int rl = (nsub == 2 && rm >= 5) ? (rm == 5 ? 2 : (rm == 6 ? 4 : 6)) : 0;  // (the kernel's RL; RM = 8 for rm = 7)
  return (size_t)oppsrc_jbuf(big, nnorb) + (size_t)T * 8 * (1 + rl);
}

// phase 2 of set_subspace (behind opp_select): can k_opp_src take the opposite-spin part of this subspace?
bool oppsrc_select(sqd_ctx* c, int64_t na, int64_t nb, const int64_t* tot) {
  const int64_t L = tot[2];  // beta single links
  if (L < 1) return false;
  if (!c->oppsrc) c->oppsrc = new OppSrcState();
  OppSrcState* s = static_cast<OppSrcState*>(c->oppsrc);
  // Geometry.  512 threads with two sub-runs (8 links, <= 128 registers), ranges of <= 512 columns, pieces of 16 entries:
  // 68 KB of LDS, two workgroups per CU that cover each other's staging round trips; rows of more than 4096 columns need
  // 1024 threads for the thread's <= OPPS_RMAX target columns (ranges of <= 1024 columns, pieces of 8 entries).
  // (3073-4096 columns with 512 threads would be seven or eight columns per thread, i.e. one sub-run: 5.4 against 3.8 ms at 4000^2)
  int T = nb <= (int64_t)5 * 512 ? 512 : 1024;  // (six columns per thread at 512 threads: 82 KB of LDS, one workgroup per CU)
  if (const char* env = std::getenv("SQD_OPPS_T")) {  // tuning / test hook (small workgroups: many passes on small sets)
    const int v = std::atoi(env);
    if (v >= 64 && v <= 1024 && v % 64 == 0) T = v;
  }
  int nsub = OPPS_NSUB_MAX;
  if (const char* env = std::getenv("SQD_OPPS_S")) {  // tuning / test hook: links per thread (rounded up to whole sub-runs)
    const int v = (std::atoi(env) + OPPS_SUB - 1) / OPPS_SUB;
    if (v >= 1 && v <= OPPS_NSUB_MAX) nsub = v;
  }
  if (nb > (int64_t)OPPS_RMAX * T) return false;
  if (c->nnorb > OppSrcLds<true>::WROWS) return false;
  bool big = T > OppSrcLds<false>::COLS || c->nnorb > OppSrcLds<false>::WROWS;
  if (const char* env = std::getenv("SQD_OPPS_BIG"))  // test hook: the 4-plane layout on small workgroups
    if (std::atoi(env) != 0) big = true;
  // (seven or eight target columns per thread beside two sub-runs: 18 spilled registers -- unless six of them live in LDS,
  // which the 1024-thread layout has room for)
  if (nb > (int64_t)6 * T && !big) nsub = 1;
  if ((size_t)OPPS_SUB * nsub * T * 8 > (size_t)(big ? OppSrcLds<true>::STAGE_BYTES : OppSrcLds<false>::STAGE_BYTES)) return false;
  if (oppsrc_shmem(big, c->nnorb, T, nsub, nb) + 1024 > (size_t)c->lds_bytes) return false;
  // a source column's links must fit one pass even if every one of them opens a sub-run of its own
  const int64_t* ps = c->h_sptr_b;
  int64_t longest = 0;
  for (int64_t B = 0; B < nb; ++B) longest = std::max(longest, ps[B + 1] - ps[B]);
  if (longest > (int64_t)nsub * T) return false;
  s->nsub = nsub;
  s->big = big;
  s->T = T;
  s->shmem = oppsrc_shmem(big, c->nnorb, T, nsub, nb);
  // work items: a row's entries (itself + its alpha single links) in pieces of at most E (<= 64: one entry per lane), so
  // that the rows of the Hartree-Fock neighbourhood (up to 177 entries) do not run as one workgroup's chain; a row in one
  // piece is written in place, the others as partial rows added in slot order.  Longest pieces first.
  int E = OPPS_EMAX;  // (64 against 32: 1-3 % -- fewer folds and partial rows per multiply-add)
  if (const char* env = std::getenv("SQD_OPPS_E")) {  // tuning / test hook (short pieces: many partial rows)
    const int v = std::atoi(env);
    if (v >= 2 && v <= OPPS_EMAX) E = v / 2 * 2;
  }
  const int64_t* pa = c->h_sptr;
  s->h_items.clear();
  s->h_multi.clear();
  s->h_rowinfo.assign((size_t)2 * na, 0);
  int32_t nslots = 0;
  for (int64_t A = 0; A < na; ++A) {
    const int nent = 1 + (int)(pa[A + 1] - pa[A]);
    const int pieces = (nent + E - 1) / E;
    if (pieces == 1) {
      s->h_items.push_back(OppSrcItem{(uint32_t)A, 0, nent, -1});
    } else {
      s->h_multi.push_back(MultiRow{(uint32_t)A, nslots, pieces});
      s->h_rowinfo[2 * A] = nslots;
      s->h_rowinfo[2 * A + 1] = pieces;
      for (int p = 0; p < pieces; ++p) {
        const int e0 = p * E, ne = (nent - e0 < E) ? nent - e0 : E;
        s->h_items.push_back(OppSrcItem{(uint32_t)A, e0, ne, nslots++});
      }
    }
  }
  std::stable_sort(s->h_items.begin(), s->h_items.end(), [](const OppSrcItem& a, const OppSrcItem& b) { return a.ne > b.ne; });
  s->n_items = (int64_t)s->h_items.size();
  s->n_slots = nslots;
  s->n_multi = (int64_t)s->h_multi.size();
  return true;
}

// The pass tables (host): the beta single links come back from the device once per subspace (8 + 4 bytes per link), are
// cut into source-column ranges of at most `cap` slots and min(T, plane) columns, grouped by widx inside a range and laid
// out thread by thread.
int oppsrc_build(sqd_ctx* c) {
  OppSrcState* s = static_cast<OppSrcState*>(c->oppsrc);
  const SpinTables& tb = c->sp[1];
  const int64_t nb = c->nb, L = c->h_sptr_b[nb];
  const int T = s->T, nsub = s->nsub, S = OPPS_SUB * nsub, nw = 2 * c->nnorb;
  s->h_rec.resize((size_t)L);
  s->h_row.resize((size_t)L);
  SQD_HIP_CHECK(hipMemcpyAsync(s->h_rec.data(), tb.s_rec.p, (size_t)L * sizeof(SRec), hipMemcpyDeviceToHost, c->stream));
  SQD_HIP_CHECK(hipMemcpyAsync(s->h_row.data(), tb.s_row.p, (size_t)L * 4, hipMemcpyDeviceToHost, c->stream));
  SQD_STREAM_SYNC(c->stream);
  // links by source column, in link order
  std::vector<int64_t> sptr((size_t)nb + 1, 0);
  for (int64_t l = 0; l < L; ++l) ++sptr[s->h_rec[l].src + 1];
  for (int64_t B = 0; B < nb; ++B) sptr[B + 1] += sptr[B];
  std::vector<uint32_t> bysrc((size_t)L);
  {
    std::vector<int64_t> fill(sptr.begin(), sptr.end() - 1);
    for (int64_t l = 0; l < L; ++l) bysrc[fill[s->h_rec[l].src]++] = (uint32_t)l;
  }
  // ranges: greedy over the source columns; slots of a range = sum over widx of its link count rounded up to sub-runs
  const int64_t cap_slots = (int64_t)S * T;
  const int cap_cols = std::min(T, s->big ? OppSrcLds<true>::COLS : OppSrcLds<false>::COLS);
  std::vector<int32_t>& cut = s->h_colcut;
  cut.assign(1, 0);
  {
    std::vector<int32_t> cnt((size_t)nw, 0);
    std::vector<int32_t> touched;
    int64_t slots = 0;
    int width = 0;
    for (int64_t B = 0; B < nb; ++B) {
      for (int attempt = 0; attempt < 2; ++attempt) {
        int64_t add = 0;
        for (int64_t i = sptr[B]; i < sptr[B + 1]; ++i) {
          const uint32_t w = srec_widx(s->h_rec[bysrc[i]].meta);
          if (cnt[w] % OPPS_SUB == 0) add += OPPS_SUB;
          if (cnt[w]++ == 0) touched.push_back((int32_t)w);
        }
        if (attempt == 1 || (slots + add <= cap_slots && width + 1 <= cap_cols)) {
          slots += add;
          ++width;
          break;
        }
        // close the range in front of B and count B again in a fresh one
        for (int32_t w : touched) cnt[w] = 0;
        touched.clear();
        cut.push_back((int32_t)B);
        slots = 0;
        width = 0;
      }
    }
    cut.push_back((int32_t)nb);
  }
  const int H = (int)cut.size() - 1;
  s->H = H;
  // tables per pass: rec[S][T] | widx[nsub][T] | colq[OPPS_RMAX][T]
  const size_t per_pass = (size_t)(S + nsub + OPPS_RMAX) * T;
  s->h_tab.assign(per_pass * H, 0u);
  std::vector<int32_t> range_of((size_t)nb);
  for (int h = 0; h < H; ++h)
    for (int32_t B = cut[h]; B < cut[h + 1]; ++B) range_of[B] = h;
  // position of every link among its pass's links in target order (= link order: the CSR is sorted by target column),
  // and per pass and column the first position | the number of its links << 16
  std::vector<uint32_t> rank((size_t)L);
  {
    std::vector<uint32_t> counter((size_t)H, 0u), first((size_t)H, 0u);
    const int64_t* ps = c->h_sptr_b;
    for (int64_t B = 0; B < nb; ++B) {
      first = counter;
      for (int64_t l = ps[B]; l < ps[B + 1]; ++l) rank[l] = counter[range_of[s->h_rec[l].src]]++;
      for (int h = 0; h < H; ++h) {
        const uint32_t n = counter[h] - first[h];
        if (n > 0xffffu || first[h] > OPPS_POS) {
          set_error("internal: opposite-spin pass tables: column run out of range");
          return SQD_ERR_STATE;
        }
        s->h_tab[per_pass * h + (size_t)(S + nsub + B / T) * T + (size_t)(B % T)] = first[h] | (n << 16);
      }
    }
  }
  {
    std::vector<std::vector<uint32_t>> by_w((size_t)nw);
    std::vector<int32_t> used;
    for (int h = 0; h < H; ++h) {
      uint32_t* rec = s->h_tab.data() + per_pass * h;
      uint32_t* wofs = rec + (size_t)S * T;
      used.clear();
      for (int32_t B = cut[h]; B < cut[h + 1]; ++B)
        for (int64_t i = sptr[B]; i < sptr[B + 1]; ++i) {
          const uint32_t l = bysrc[i], w = srec_widx(s->h_rec[l].meta);
          if (by_w[w].empty()) used.push_back((int32_t)w);
          by_w[w].push_back(l);
        }
      std::sort(used.begin(), used.end());
      int64_t u = 0;  // sub-run index: thread u % T, sub-run u / T of that thread
      for (int32_t w : used) {
        std::vector<uint32_t>& ls = by_w[w];
        std::sort(ls.begin(), ls.end());
        for (size_t i0 = 0; i0 < ls.size(); i0 += OPPS_SUB, ++u) {
          const int t = (int)(u % T), j = (int)(u / T);
          if (j >= nsub) {
            set_error("internal: opposite-spin pass tables overflow");
            return SQD_ERR_STATE;
          }
          wofs[(size_t)j * T + t] = (uint32_t)w;
          for (int k = 0; k < OPPS_SUB && i0 + k < ls.size(); ++k) {
            const uint32_t l = ls[i0 + k];
            const size_t at = (size_t)(OPPS_SUB * j + k) * T + t;
            rec[at] = ((uint32_t)(s->h_rec[l].src - (uint32_t)cut[h]) * 16u) | (rank[l] << OPPS_POS_SHIFT) |
                      ((s->h_rec[l].meta >> 31) ? OPPS_NEG : 0u) | OPPS_LIVE;
          }
        }
        ls.clear();
      }
    }
  }
  SQD_TRY(s->tab.reserve(s->h_tab.size() * 4 + 64));
  SQD_TRY(s->colcut.reserve(s->h_colcut.size() * 4 + 64));
  SQD_TRY(s->items.reserve((size_t)s->n_items * sizeof(OppSrcItem) + 64));
  SQD_TRY(s->rowinfo.reserve((size_t)2 * c->na * 4 + 64));
  SQD_TRY(s->multi.reserve((size_t)s->n_multi * sizeof(MultiRow) + 64));
  SQD_TRY(s->partial.reserve((size_t)s->n_slots * c->nb * 8 + 64));
  SQD_HIP_CHECK(hipMemcpyAsync(s->tab.p, s->h_tab.data(), s->h_tab.size() * 4, hipMemcpyHostToDevice, c->stream));
  SQD_HIP_CHECK(hipMemcpyAsync(s->colcut.p, s->h_colcut.data(), s->h_colcut.size() * 4, hipMemcpyHostToDevice, c->stream));
  SQD_HIP_CHECK(hipMemcpyAsync(s->items.p, s->h_items.data(), (size_t)s->n_items * sizeof(OppSrcItem), hipMemcpyHostToDevice, c->stream));
  SQD_HIP_CHECK(hipMemcpyAsync(s->rowinfo.p, s->h_rowinfo.data(), (size_t)2 * c->na * 4, hipMemcpyHostToDevice, c->stream));
  if (s->n_multi)
    SQD_HIP_CHECK(hipMemcpyAsync(s->multi.p, s->h_multi.data(), (size_t)s->n_multi * sizeof(MultiRow), hipMemcpyHostToDevice, c->stream));
  return SQD_OK;
}

// number of source-column passes of the latest build (probes / tests)
int oppsrc_passes(const sqd_ctx* c) {
  const OppSrcState* s = static_cast<const OppSrcState*>(c->oppsrc);
  return s ? s->H : 0;
}

// sigma = (hdiag + opposite-spin part) c + G, G = sqd_ctx::gdense as spmm_launch has just formed it for the same vector
int oppsrc_launch(sqd_ctx* c, const double* d_c, double* d_sigma, int64_t in_stride, int64_t out_stride, bool spin, double ss,
               double shift) {
  OppSrcState* s = static_cast<OppSrcState*>(c->oppsrc);
  if (!s) {
    set_error("internal: opposite-spin row kernel without its tables");
    return SQD_ERR_STATE;
  }
  OppSrcArgs g;
  const SpinTables& ta = c->sp[0];
  const SpinTables& tb = c->sp[1];
  g.c = d_c;
  g.sigma = d_sigma;
  g.hdiag = c->hdiag.as<double>();
  g.gdense = c->gdense.as<double>();
  g.ja_row = ta.jrow.as<double>();
  g.jbT = tb.jT.as<double>();
  g.eri_pp = c->eri_pp.as<double>();
  g.sa_ptr = ta.s_ptr.as<int64_t>();
  g.sa_rec = ta.s_rec.as<SRec>();
  g.tab = s->tab.as<uint32_t>();
  g.colcut = s->colcut.as<int32_t>();
  g.items = s->items.as<OppSrcItem>();
  g.partial = s->partial.as<double>();
  g.nb = c->nb;
  g.nnorb = c->nnorb;
  g.T = s->T;
  g.H = s->H;
  g.jbuf_off = oppsrc_jbuf(s->big, c->nnorb);
  g.stop = c->sigma_stop;
  const bool indexed = c->sigma_index && (in_stride || out_stride);
  g.vec_index = indexed ? c->sigma_index : nullptr;
  g.c_stride = in_stride;
  g.s_stride = out_stride;
  g.spin = spin ? 1 : 0;
  g.ss = ss;
  g.shift = shift;
  {
    const double sz = 0.5 * (c->nelec[0] - c->nelec[1]);
    g.szterm = sz * (sz + 1.0);
  }
  g.strs_a = c->sp[0].strs.as<uint64_t>();
  g.strs_b = c->sp[1].strs.as<uint64_t>();
  g.n_items = (unsigned)s->n_items;
  const int rm = (int)((c->nb + s->T - 1) / s->T);  // target columns per thread (<= OPPS_RMAX: oppsrc_select)
  const dim3 grid((unsigned)s->n_items), block((unsigned)s->T);
#define SQD_OPPS_CASE(NSUB_, RM_, BIG_)                                                                             \
  do {                                                                                                             \
    if (s->shmem > 64 * 1024) {                                                                                    \
      static std::atomic<size_t> granted[64];                                                                      \
      const int dev = c->device & 63;                                                                              \
      if (s->shmem > granted[dev].load(std::memory_order_relaxed)) {                                               \
        SQD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_opp_src<NSUB_, RM_, BIG_>),            \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->shmem));             \
        granted[dev].store(s->shmem, std::memory_order_relaxed);                                                   \
      }                                                                                                            \
    }                                                                                                              \
    hipLaunchKernelGGL((k_opp_src<NSUB_, RM_, BIG_>), grid, block, s->shmem, c->stream, g);                       \
  } while (0)
#define SQD_OPPS_BIG(NSUB_, RM_)                        \
  do {                                                 \
    if (s->big) SQD_OPPS_CASE(NSUB_, RM_, true);        \
    else SQD_OPPS_CASE(NSUB_, RM_, false);              \
  } while (0)
  if (s->nsub == 2 && rm > 6) {  // (1024 threads only: oppsrc_select)
    SQD_OPPS_CASE(2, 8, true);
  } else if (s->nsub == 1 || rm > 6) {  // (512 threads, rows of more than six columns per thread: one sub-run, oppsrc_select)
    if (rm <= 2) SQD_OPPS_BIG(1, 2);
    else if (rm <= 4) SQD_OPPS_BIG(1, 4);
    else if (rm <= 6) SQD_OPPS_BIG(1, 6);
    else SQD_OPPS_BIG(1, 8);
  } else {
    if (rm <= 2) SQD_OPPS_BIG(2, 2);
    else if (rm <= 4) SQD_OPPS_BIG(2, 4);
    else if (rm == 5) SQD_OPPS_BIG(2, 5);
    else SQD_OPPS_BIG(2, 6);
  }
#undef SQD_OPPS_BIG
#undef SQD_OPPS_CASE
  SQD_HIP_CHECK(hipGetLastError());
  // rows in several pieces: inside a Davidson run the first reader of the new vector adds the partial rows (oppsrc_split)
  if (s->n_multi > 0 && !(c->sigma_defer_reduce && indexed)) {
    OppSrcReduceArgs r;
    r.rows = s->multi.as<MultiRow>();
    r.partial = s->partial.as<double>();
    r.sigma = d_sigma;
    r.nb = c->nb;
    r.stop = g.stop;
    r.vec_index = g.vec_index;
    r.s_stride = out_stride;
    hipLaunchKernelGGL(k_opp_src_reduce, dim3((unsigned)s->n_multi, (unsigned)((c->nb + 1023) / 1024)), dim3(256), 0, c->stream, r);
    SQD_HIP_CHECK(hipGetLastError());
  }
  if (c->ev_after_sigma_kernel) {
    SQD_HIP_CHECK(hipEventRecord(c->ev_after_sigma_kernel, c->stream));
    c->ev_after_sigma_kernel = nullptr;
  }
  return SQD_OK;
}

// the split-row records of the latest oppsrc_select (for k_dots_eig's deferred sum); false: every row is in one piece
bool oppsrc_split(const sqd_ctx* c, const int32_t** rowinfo, const double** partial) {
  const OppSrcState* s = static_cast<const OppSrcState*>(c->oppsrc);
  if (!s || s->n_multi == 0) return false;
  *rowinfo = s->rowinfo.as<int32_t>();
  *partial = s->partial.as<double>();
  return true;
}

}  // namespace sqd

#ifdef SQD_PHASE_CLOCK
// out[8]: column sums over the workgroups' rows {passes, stage, gather, fold 1, fold 2, whole item, items, rounds} (10 ns ticks)
extern "C" __attribute__((visibility("default"))) int sqd_probe_clk_oppsrc(unsigned long long* out, int reset) {
  static std::vector<unsigned long long> h((size_t)sqd::OCLK_ROWS * sqd::OCLK_COLS);
  if (out) {
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(sqd::sqd_clk_oppsrc), h.size() * 8) != hipSuccess) return -1;
    for (int k = 0; k < sqd::OCLK_COLS; ++k) out[k] = 0;
    for (size_t r = 0; r < (size_t)sqd::OCLK_ROWS; ++r)
      for (int k = 0; k < sqd::OCLK_COLS; ++k) out[k] += h[r * sqd::OCLK_COLS + k];
  }
  if (reset) {
    std::fill(h.begin(), h.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(sqd::sqd_clk_oppsrc), h.data(), h.size() * 8) != hipSuccess) return -1;
  }
  return 0;
}
#endif
