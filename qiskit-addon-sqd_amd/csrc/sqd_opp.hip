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
// Round 6 formulation (k_opp_rows; the round-5 kernel of the same name kept the beta links by TARGET column range, every
// range staging whole source rows: 4.7 GB per sigma at 3000 x 3000 out of the Infinity Cache, two LDS gathers per
// multiply-add, 6-40 spilled registers on rows of more than 4096 columns).  ONE workgroup owns a piece of a target row A
// (<= E of its entries) and walks the beta link list in PASSES over ranges of the SOURCE column B':
//   * a pass stages only its range [q0, q1) of every source row -- each element of a source row is staged once per item,
//     not once per range -- two entries at a time, interleaved (Cst[B' - q0][2], signed) and double-buffered in LDS;
//   * inside a range the links are grouped by excitation operator (widx = orbital pair and direction) in sub-runs of four:
//     a thread holds <= NSUB sub-runs -- per link ONE register, the byte offset of its source column in Cst, and one
//     accumulator -- and per sub-run the byte offset of its weight pair in Wst[widx][2]: a link costs one 16-byte LDS
//     gather and two multiply-adds, a sub-run one more gather; no address arithmetic, no record decoding, and the linear
//     spin penalty is one addition to a staged weight (Wst[partner of the alpha link][entry] -= shift);
//   * the alpha single x beta occupation term rides on the staging pass (the staged columns' J values in registers);
//   * at the end of a pass the per-link sums go through LDS to the threads that own the target columns (positions in
//     target order precomputed per range; runs summed in that order), and a thread carries its columns' sums over the
//     passes in registers: one sigma row per item, written once, same bits on every run.
// Rows in one piece are written in place; the others as partial rows that the first reader of the vector adds in slot
// order (k_dots_s inside a Davidson run, k_opp_reduce otherwise).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "sqd_common.h"

namespace sqd {

constexpr int OPP_SUB = 4;       // links per sub-run (one weight gather serves four links)
constexpr int OPP_NSUB_MAX = 2;  // sub-runs per thread (8 links: 8 + 16 registers; three sub-runs spill at 1024 threads' 128 registers)
constexpr int OPP_JR_MAX = 2;    // staged columns per thread and pass (a range is <= JR * threads columns wide)
constexpr int OPP_RMAX = 8;      // target columns per thread (nb <= OPP_RMAX * threads)
// LDS plan (bytes; compile-time offsets so that the buffer of a batch is an immediate of the gather instruction):
//   Cst[2][COLS][2] | Wst[2][ROWS][2] -- the staging buffers; accb[links of a pass] takes their place at the end of a pass
//   and jbuf[JR * T] sits behind.  BIG = false: 64 + 8 KB, two workgroups of 512 threads per CU (norb <= 31);
//   BIG = true: 128 + 16 KB (norb <= 44, or 1024 threads)
template <bool BIG>
struct OppLds {
  static constexpr int COLS = BIG ? 2048 : 1024, ROWS = BIG ? 2048 : 1024;
  static constexpr int CST0 = 0, CST1 = COLS * 16, WST0 = 2 * COLS * 16, WST1 = WST0 + ROWS * 16;
  static constexpr int STAGE_BYTES = WST1 + ROWS * 16;
  static constexpr int JBUF = STAGE_BYTES;
};
constexpr uint32_t OPP_DEAD = 0xffffffffu;

// one workgroup's share of a row: entries [e0, e0 + ne) of row A (entry 0 = the row itself, entry e > 0 = its alpha single
// link e - 1); slot < 0: the row has this one item and is written in place, else partial row `slot` (added in slot
// order by the first reader of the vector -- k_dots_eig inside a Davidson run, k_opp_reduce otherwise)
struct OppItem {
  uint32_t A;
  int32_t e0, ne, slot;
};
struct OppState {
  DevBuf tab, cptr, colcut, items, rowinfo, partial, multi;
  std::vector<OppItem> h_items;
  std::vector<int32_t> h_rowinfo;
  std::vector<MultiRow> h_multi;
  std::vector<uint32_t> h_tab, h_cptr;
  std::vector<int32_t> h_colcut;
  std::vector<SRec> h_rec;
  std::vector<uint32_t> h_row;
  int H = 1, nsub = 2, jr = 1, T = 512;
  bool big = false;
  int64_t n_items = 0, n_slots = 0, n_multi = 0, n_slots_used = 0;
  size_t shmem = 0;
};

void opp_release(sqd_ctx* c) {
  if (!c->opp) return;
  OppState* s = static_cast<OppState*>(c->opp);
  for (DevBuf* b : {&s->tab, &s->cptr, &s->colcut, &s->items, &s->rowinfo, &s->partial, &s->multi}) b->release();
  delete s;
  c->opp = nullptr;
}

struct OppArgs {
  GPtr<const double> c;
  GPtr<double> sigma, partial;
  GPtr<const double> hdiag, gdense, ja_row, jbT, eri_pp;
  GPtr<const int64_t> sa_ptr;
  GPtr<const SRec> sa_rec;
  GPtr<const uint32_t> tab;    // per pass: rec[S][T] | wofs[NSUB][T] | pos[S][T]
  GPtr<const uint32_t> cptr;   // [H][nb + 1] first position (target order) of every column's links inside the pass
  GPtr<const int32_t> colcut;  // [H + 1] first source column of every pass
  GPtr<const OppItem> items;
  int64_t nb;
  int nnorb, T, H;
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

// Values every lane of the workgroup agrees on, pinned to scalar registers.  (Loads behind a barrier are vector loads to
// the compiler -- the fence in __syncthreads() counts as a clobber -- and everything derived from them, row pointers
// included, would live in vector registers: 64-bit addresses per load instead of a scalar base + 32-bit offset.)
__device__ inline int opp_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ inline uint32_t opp_uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline int64_t opp_uni(int64_t v) {
  const uint32_t lo = opp_uni((uint32_t)v), hi = opp_uni((uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

template <int NSUB, int JR, int RM, bool BIG>
__global__ void __launch_bounds__(1024) k_opp_rows(const OppArgs g) {
  constexpr int S = OPP_SUB * NSUB;
  using Lds = OppLds<BIG>;
  HIP_DYNAMIC_SHARED(double, smem)
  char* const lds = reinterpret_cast<char*>(smem);
  if (g.stop && *g.stop) return;
  const unsigned item_index = blockIdx.x;
  if (item_index >= g.n_items) return;
  const int T = g.T, tid = threadIdx.x;
  OppItem it = g.items[item_index];
  it.A = opp_uni(it.A);
  it.e0 = opp_uni(it.e0);
  it.ne = opp_uni(it.ne);
  it.slot = opp_uni(it.slot);
  const int64_t A = it.A;
  const int64_t nb = g.nb;
  const int nn = g.nnorb, nw = 2 * g.nnorb;
  const int64_t vsel = g.vec_index ? (int64_t)opp_uni(*g.vec_index - 1) : 0;
  const double* __restrict__ C = g.c + vsel * g.c_stride;
  double* __restrict__ sig = g.sigma + vsel * g.s_stride;
  const int64_t k0 = opp_uni(g.sa_ptr[A]);
  const int e_end = it.e0 + it.ne;
  const bool spin = g.spin != 0;
  const double pen = -g.shift;
  double colacc[RM];
#pragma unroll
  for (int r = 0; r < RM; ++r) colacc[r] = 0.0;
  double* const accb = smem;
  double* const jbuf = reinterpret_cast<double*>(lds + Lds::JBUF);

  for (int h = 0; h < g.H; ++h) {
    const int q0 = opp_uni(g.colcut[h]), q1 = opp_uni(g.colcut[h + 1]);
    const uint32_t* __restrict__ tab = g.tab + (int64_t)h * ((2 * S + NSUB) * (int64_t)T) + tid;
    uint32_t rec[S], wofs[NSUB];
#pragma unroll
    for (int s = 0; s < S; ++s) rec[s] = tab[(int64_t)s * T];
#pragma unroll
    for (int j = 0; j < NSUB; ++j) wofs[j] = tab[(int64_t)(S + j) * T];
    double acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = 0.0;
    double jacc[JR];
#pragma unroll
    for (int i = 0; i < JR; ++i) jacc[i] = 0.0;

    // Register-staged prefetch: the global loads of batch b + 1 (the range's share of two source rows and two J rows, two
    // weight values) are requested right behind the barrier that publishes batch b and land while batch b is gathered.
    double px[JR][2], pjb[JR][2], pw[2], psg[2];
    bool plnk[2];
    int ppart[2];
    const double* pwrow[2];
    auto request = [&](int e0) {
      const double* srow[2];
      const double* jrow[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int e = e0 + j;
        const bool valid = e < e_end;
        plnk[j] = valid && e > 0;
        SRec r = SRec{(uint32_t)A, 0u};
        if (plnk[j]) r = g.sa_rec[k0 + e - 1];
        r.src = opp_uni(r.src);
        r.meta = opp_uni(r.meta);
        const uint32_t widx = srec_widx(r.meta);
        const int64_t pair = (int64_t)(widx >> 1);
        ppart[j] = plnk[j] ? (int)(widx ^ 1u) : -1;  // S^2: same orbital pair, opposite direction
        srow[j] = C + (int64_t)r.src * nb;
        psg[j] = valid ? (plnk[j] ? srec_sign(r.meta) : 1.0) : 0.0;
        pwrow[j] = plnk[j] ? g.eri_pp + pair * nn : g.ja_row + A * nn;
        jrow[j] = g.jbT + pair * nb;
      }
      const int iw = (tid < nw ? tid : nw - 1) >> 1;
#pragma unroll
      for (int j = 0; j < 2; ++j) pw[j] = pwrow[j][iw];
#pragma unroll
      for (int i = 0; i < JR; ++i) {
        const int B = q0 + tid + i * T;
        const int Bc = B < q1 ? B : q1 - 1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          px[i][j] = srow[j][Bc];
          pjb[i][j] = jrow[j][Bc];
        }
      }
    };
    // registers -> LDS buffer BUF (signed source values and weights, interleaved); the alpha single x beta occupation
    // term of the staged columns on the way
    auto park = [&](int cst, int wst) {
      for (int w = tid; w < nw; w += T) {
        double v[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          v[j] = (w == tid) ? pw[j] : pwrow[j][w >> 1];
          v[j] = psg[j] != 0.0 ? v[j] : 0.0;
          if (spin && w == ppart[j]) v[j] += pen;
        }
        *reinterpret_cast<double2*>(lds + wst + w * 16) = make_double2(v[0], v[1]);
      }
#pragma unroll
      for (int i = 0; i < JR; ++i) {
        const int Bl = tid + i * T;
        if (q0 + Bl < q1) {
          const double x0 = px[i][0] * psg[0], x1 = px[i][1] * psg[1];
          jacc[i] += plnk[0] ? pjb[i][0] * x0 : 0.0;
          jacc[i] += plnk[1] ? pjb[i][1] * x1 : 0.0;
          *reinterpret_cast<double2*>(lds + cst + Bl * 16) = make_double2(x0, x1);
        }
      }
    };
    // (sub-run j + 1's five reads are issued before sub-run j's eight multiply-adds; the scheduling barriers keep it at two
    // sub-runs -- 40 registers -- in flight: left alone the compiler hoists all 5 NSUB reads and spills the accumulators)
    auto gather = [&](int cst, int wst) {
      double2 w2[2], c2[2][OPP_SUB];
      auto read = [&](int j) {
        w2[j & 1] = *reinterpret_cast<const double2*>(lds + wst + wofs[j]);
#pragma unroll
        for (int k = 0; k < OPP_SUB; ++k) c2[j & 1][k] = *reinterpret_cast<const double2*>(lds + cst + rec[OPP_SUB * j + k]);
      };
      read(0);
#pragma unroll
      for (int j = 0; j < NSUB; ++j) {
        if (j + 1 < NSUB) read(j + 1);
#pragma unroll
        for (int k = 0; k < OPP_SUB; ++k) {
          acc[OPP_SUB * j + k] += w2[j & 1].x * c2[j & 1][k].x;
          acc[OPP_SUB * j + k] += w2[j & 1].y * c2[j & 1][k].y;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    request(it.e0);
    for (int e0 = it.e0; e0 < e_end; e0 += 4) {
      park(Lds::CST0, Lds::WST0);
      __syncthreads();
      if (e0 + 2 < e_end) request(e0 + 2);
      gather(Lds::CST0, Lds::WST0);
      if (e0 + 2 < e_end) {
        park(Lds::CST1, Lds::WST1);
        __syncthreads();
        if (e0 + 4 < e_end) request(e0 + 4);
        gather(Lds::CST1, Lds::WST1);
      }
    }
    __syncthreads();  // every gather of the pass is done: the staging buffers become accb
    // ---- per-link sums -> target columns.  pos = the link's position among the pass's links in target order (sign of
    // the beta link in bit 31); the owner of a column adds its run in that order, then the staged columns' J term.
    const uint32_t* __restrict__ ptab = tab + (int64_t)(S + NSUB) * T;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const uint32_t p = ptab[(int64_t)s * T];
      if (p != OPP_DEAD) accb[p & 0x7fffffffu] = (p >> 31) ? -acc[s] : acc[s];
    }
#pragma unroll
    for (int i = 0; i < JR; ++i) jbuf[tid + i * T] = jacc[i];
    __syncthreads();
    const uint32_t* __restrict__ cp = g.cptr + (int64_t)h * (nb + 1);
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int64_t B = tid + (int64_t)r * T;
      if (B < nb) {
        const uint32_t c0 = cp[B], c1 = cp[B + 1];
        double sum = 0.0;
        for (uint32_t i = c0; i < c1; ++i) sum += accb[i];
        if (B >= q0 && B < q1) sum += jbuf[B - q0];
        colacc[r] += sum;
      }
    }
    __syncthreads();  // (accb / jbuf are read: the next pass may stage)
  }
  const bool has0 = it.e0 == 0;  // the piece that holds the row itself also brings the diagonal and the same-spin product
  const double* __restrict__ crow = C + A * nb;
  const double* __restrict__ hd = g.hdiag + A * nb;
  const double* __restrict__ gd = g.gdense + A * nb;
  double* __restrict__ orow = it.slot < 0 ? sig + A * nb : g.partial + (int64_t)it.slot * nb;
#pragma unroll
  for (int r = 0; r < RM; ++r) {
    const int64_t B = tid + (int64_t)r * T;
    if (B < nb) {
      double v = colacc[r];
      if (has0) {
        double d = hd[B];
        if (spin) d += g.shift * (g.szterm + (double)__popcll(g.strs_b[B] & ~g.strs_a[A]) - g.ss);
        v += d * crow[B] + gd[B];
      }
      orow[B] = v;
    }
  }
}

// sigma[A, :] = sum of the partial rows of A in slot order, for the rows that were cut into several items (outside
// Davidson runs; inside, k_dots_eig adds them as the first reader of the vector)
struct OppReduceArgs {
  GPtr<const MultiRow> rows;
  GPtr<const double> partial;
  GPtr<double> sigma;
  int64_t nb;
  GPtr<const int> stop, vec_index;
  int64_t s_stride;
};
__global__ void __launch_bounds__(256) k_opp_reduce(const OppReduceArgs g) {
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
static size_t opp_shmem(bool big, int jr, int T) {  // the staging buffers (accb in their place at the end of a pass), then jbuf
  return (size_t)(big ? OppLds<true>::JBUF : OppLds<false>::JBUF) + (size_t)jr * T * 8;
}

// phase 2 of set_subspace, behind spmm_select: is the opposite-spin part of this subspace taken by k_opp_rows?
// (SQD_SIGMA_OPP=0 forbids: the work items then add G as they add the matrix-core product.)
bool opp_select(sqd_ctx* c, int64_t na, int64_t nb, const int64_t* tot) {
  c->sig_opp = false;
  if (!c->sig_spmm) return false;
  if (const char* env = std::getenv("SQD_SIGMA_OPP"))
    if (std::atoi(env) == 0) return false;
  const int64_t L = tot[2];  // beta single links
  if (L < 1 || 2 * c->nnorb > OppLds<true>::ROWS) return false;
  if (!c->opp) c->opp = new OppState();
  OppState* s = static_cast<OppState*>(c->opp);
  // Geometry.  512 threads with two sub-runs (8 links, 128 registers) and ranges of <= 512 columns: 72 KB of LDS, two
  // workgroups per CU that cover each other's barriers; rows of more than 4096 columns need 1024 threads for the
  // thread's <= OPP_RMAX target columns.
  int T = nb <= (int64_t)OPP_RMAX * 512 ? 512 : 1024;
  if (const char* env = std::getenv("SQD_OPP_T")) {  // tuning / test hook (small workgroups: many passes on small sets)
    const int v = std::atoi(env);
    if (v >= 64 && v <= 1024 && v % 64 == 0) T = v;
  }
  int nsub = 2;
  if (const char* env = std::getenv("SQD_OPP_S")) {  // tuning / test hook: links per thread (rounded up to whole sub-runs)
    const int v = (std::atoi(env) + OPP_SUB - 1) / OPP_SUB;
    if (v >= 1 && v <= OPP_NSUB_MAX) nsub = v;
  }
  int jr = 1;
  if (const char* env = std::getenv("SQD_OPP_JR")) {  // tuning hook: staged columns per thread and pass
    const int v = std::atoi(env);
    if (v >= 1 && v <= OPP_JR_MAX) jr = v;
  }
  if (nb > (int64_t)OPP_RMAX * T) return false;
  if (nb > (int64_t)4 * T) jr = 1;  // (two staged columns beside eight target columns per thread do not fit the registers)
  bool big = 2 * c->nnorb > OppLds<false>::ROWS || jr * T > OppLds<false>::COLS ||
             (size_t)OPP_SUB * nsub * T * 8 > (size_t)OppLds<false>::STAGE_BYTES;
  if (const char* env = std::getenv("SQD_OPP_BIG"))
    if (std::atoi(env) != 0) big = true;
  if ((size_t)OPP_SUB * nsub * T * 8 > (size_t)(big ? OppLds<true>::STAGE_BYTES : OppLds<false>::STAGE_BYTES)) return false;
  if (opp_shmem(big, jr, T) + 1024 > (size_t)c->lds_bytes) return false;
  // a source column's links must fit one pass even if every one of them opens a sub-run of its own
  const int64_t* ps = c->h_sptr_b;
  int64_t longest = 0;
  for (int64_t B = 0; B < nb; ++B) longest = std::max(longest, ps[B + 1] - ps[B]);
  if (longest > (int64_t)nsub * T) return false;
  s->nsub = nsub;
  s->jr = jr;
  s->big = big;
  s->T = T;
  s->shmem = opp_shmem(big, jr, T);
  // work items: a row's entries (itself + its alpha single links) in pieces of at most E, so that the rows of the
  // Hartree-Fock neighbourhood (up to 177 entries) do not run as one workgroup's chain of 90 batches; a row in one
  // piece is written in place, the others as partial rows added in slot order.  Longest pieces first.
  int E = 32;
  if (const char* env = std::getenv("SQD_OPP_E")) {  // tuning hook
    const int v = std::atoi(env);
    if (v >= 2 && v <= 4096) E = v / 2 * 2;
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
      s->h_items.push_back(OppItem{(uint32_t)A, 0, nent, -1});
    } else {
      s->h_multi.push_back(MultiRow{(uint32_t)A, nslots, pieces});
      s->h_rowinfo[2 * A] = nslots;
      s->h_rowinfo[2 * A + 1] = pieces;
      for (int p = 0; p < pieces; ++p) {
        const int e0 = p * E, ne = (nent - e0 < E) ? nent - e0 : E;
        s->h_items.push_back(OppItem{(uint32_t)A, e0, ne, nslots++});
      }
    }
  }
  std::stable_sort(s->h_items.begin(), s->h_items.end(), [](const OppItem& a, const OppItem& b) { return a.ne > b.ne; });
  s->n_items = (int64_t)s->h_items.size();
  s->n_slots = nslots;
  s->n_multi = (int64_t)s->h_multi.size();
  c->sig_opp = true;
  return true;
}

// The pass tables (host): the beta single links come back from the device once per subspace (8 + 4 bytes per link), are
// cut into source-column ranges of at most `cap` slots and jr * T columns, grouped by widx inside a range and laid
// out thread by thread.
int opp_build(sqd_ctx* c) {
  OppState* s = static_cast<OppState*>(c->opp);
  const SpinTables& tb = c->sp[1];
  const int64_t nb = c->nb, L = c->h_sptr_b[nb];
  const int T = s->T, nsub = s->nsub, S = OPP_SUB * nsub, nw = 2 * c->nnorb;
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
  const int cap_cols = std::min(s->jr * T, s->big ? OppLds<true>::COLS : OppLds<false>::COLS);
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
          if (cnt[w] % OPP_SUB == 0) add += OPP_SUB;
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
  // tables per pass: rec[S][T] | wofs[nsub][T] | pos[S][T];  cptr[H][nb + 1]
  const size_t per_pass = (size_t)(2 * S + nsub) * T;
  s->h_tab.assign(per_pass * H, 0u);
  s->h_cptr.assign((size_t)H * (nb + 1), 0u);
  std::vector<int32_t> range_of((size_t)nb);
  for (int h = 0; h < H; ++h)
    for (int32_t B = cut[h]; B < cut[h + 1]; ++B) range_of[B] = h;
  // position of every link among its pass's links in target order (= link order: the CSR is sorted by target column)
  std::vector<uint32_t> rank((size_t)L);
  {
    std::vector<uint32_t> counter((size_t)H, 0u);
    const int64_t* ps = c->h_sptr_b;
    for (int64_t B = 0; B < nb; ++B) {
      for (int h = 0; h < H; ++h) s->h_cptr[(size_t)h * (nb + 1) + B] = counter[h];
      for (int64_t l = ps[B]; l < ps[B + 1]; ++l) rank[l] = counter[range_of[s->h_rec[l].src]]++;
    }
    for (int h = 0; h < H; ++h) s->h_cptr[(size_t)h * (nb + 1) + nb] = counter[h];
  }
  {
    std::vector<std::vector<uint32_t>> by_w((size_t)nw);
    std::vector<int32_t> used;
    for (int h = 0; h < H; ++h) {
      uint32_t* rec = s->h_tab.data() + per_pass * h;
      uint32_t* wofs = rec + (size_t)S * T;
      uint32_t* pos = wofs + (size_t)nsub * T;
      std::fill(pos, pos + (size_t)S * T, OPP_DEAD);
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
        for (size_t i0 = 0; i0 < ls.size(); i0 += OPP_SUB, ++u) {
          const int t = (int)(u % T), j = (int)(u / T);
          if (j >= nsub) {
            set_error("internal: opposite-spin pass tables overflow");
            return SQD_ERR_STATE;
          }
          wofs[(size_t)j * T + t] = (uint32_t)w * 16u;
          for (int k = 0; k < OPP_SUB && i0 + k < ls.size(); ++k) {
            const uint32_t l = ls[i0 + k];
            const size_t at = (size_t)(OPP_SUB * j + k) * T + t;
            rec[at] = (uint32_t)(s->h_rec[l].src - (uint32_t)cut[h]) * 16u;
            pos[at] = rank[l] | ((s->h_rec[l].meta >> 31) ? 0x80000000u : 0u);
          }
        }
        ls.clear();
      }
    }
  }
  SQD_TRY(s->tab.reserve(s->h_tab.size() * 4 + 64));
  SQD_TRY(s->cptr.reserve(s->h_cptr.size() * 4 + 64));
  SQD_TRY(s->colcut.reserve(s->h_colcut.size() * 4 + 64));
  SQD_TRY(s->items.reserve((size_t)s->n_items * sizeof(OppItem) + 64));
  SQD_TRY(s->rowinfo.reserve((size_t)2 * c->na * 4 + 64));
  SQD_TRY(s->multi.reserve((size_t)s->n_multi * sizeof(MultiRow) + 64));
  SQD_TRY(s->partial.reserve((size_t)s->n_slots * c->nb * 8 + 64));
  SQD_HIP_CHECK(hipMemcpyAsync(s->tab.p, s->h_tab.data(), s->h_tab.size() * 4, hipMemcpyHostToDevice, c->stream));
  SQD_HIP_CHECK(hipMemcpyAsync(s->cptr.p, s->h_cptr.data(), s->h_cptr.size() * 4, hipMemcpyHostToDevice, c->stream));
  SQD_HIP_CHECK(hipMemcpyAsync(s->colcut.p, s->h_colcut.data(), s->h_colcut.size() * 4, hipMemcpyHostToDevice, c->stream));
  SQD_HIP_CHECK(hipMemcpyAsync(s->items.p, s->h_items.data(), (size_t)s->n_items * sizeof(OppItem), hipMemcpyHostToDevice, c->stream));
  SQD_HIP_CHECK(hipMemcpyAsync(s->rowinfo.p, s->h_rowinfo.data(), (size_t)2 * c->na * 4, hipMemcpyHostToDevice, c->stream));
  if (s->n_multi)
    SQD_HIP_CHECK(hipMemcpyAsync(s->multi.p, s->h_multi.data(), (size_t)s->n_multi * sizeof(MultiRow), hipMemcpyHostToDevice, c->stream));
  return SQD_OK;
}

// number of source-column passes of the latest build (probes / tests)
int opp_passes(const sqd_ctx* c) {
  const OppState* s = static_cast<const OppState*>(c->opp);
  return (s && c->sig_opp) ? s->H : 0;
}

// sigma = (hdiag + opposite-spin part) c + G, G = sqd_ctx::gdense as spmm_launch has just formed it for the same vector
int opp_launch(sqd_ctx* c, const double* d_c, double* d_sigma, int64_t in_stride, int64_t out_stride, bool spin, double ss,
               double shift) {
  OppState* s = static_cast<OppState*>(c->opp);
  if (!s) {
    set_error("internal: opposite-spin row kernel without its tables");
    return SQD_ERR_STATE;
  }
  OppArgs g;
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
  g.cptr = s->cptr.as<uint32_t>();
  g.colcut = s->colcut.as<int32_t>();
  g.items = s->items.as<OppItem>();
  g.partial = s->partial.as<double>();
  g.nb = c->nb;
  g.nnorb = c->nnorb;
  g.T = s->T;
  g.H = s->H;
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
  const int rm = (int)((c->nb + s->T - 1) / s->T);  // target columns per thread (<= OPP_RMAX: opp_select)
  const dim3 grid((unsigned)s->n_items), block((unsigned)s->T);
#define SQD_OPP_CASE(NSUB_, JR_, RM_, BIG_)                                                                        \
  do {                                                                                                             \
    if (s->shmem > 64 * 1024) {                                                                                    \
      static std::atomic<size_t> granted[64];                                                                      \
      const int dev = c->device & 63;                                                                              \
      if (s->shmem > granted[dev].load(std::memory_order_relaxed)) {                                               \
        SQD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_opp_rows<NSUB_, JR_, RM_, BIG_>),       \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->shmem));             \
        granted[dev].store(s->shmem, std::memory_order_relaxed);                                                   \
      }                                                                                                            \
    }                                                                                                              \
    hipLaunchKernelGGL((k_opp_rows<NSUB_, JR_, RM_, BIG_>), grid, block, s->shmem, c->stream, g);                  \
  } while (0)
#define SQD_OPP_BIG(NSUB_, JR_, RM_)                        \
  do {                                                      \
    if (s->big) SQD_OPP_CASE(NSUB_, JR_, RM_, true);        \
    else SQD_OPP_CASE(NSUB_, JR_, RM_, false);              \
  } while (0)
#define SQD_OPP_RM(NSUB_, JR_)                     \
  do {                                             \
    if (rm <= 2) SQD_OPP_BIG(NSUB_, JR_, 2);       \
    else if (rm <= 4) SQD_OPP_BIG(NSUB_, JR_, 4);  \
    else SQD_OPP_BIG(NSUB_, JR_, 8);               \
  } while (0)
#define SQD_OPP_JR(NSUB_)                                 \
  do {                                                    \
    if (s->jr == 1) SQD_OPP_RM(NSUB_, 1);                 \
    else if (rm <= 2) SQD_OPP_BIG(NSUB_, 2, 2);           \
    else SQD_OPP_BIG(NSUB_, 2, 4); /* (rm <= 4: opp_select) */ \
  } while (0)
  if (s->nsub == 1) SQD_OPP_JR(1);
  else SQD_OPP_JR(2);
#undef SQD_OPP_JR
#undef SQD_OPP_RM
#undef SQD_OPP_BIG
#undef SQD_OPP_CASE
  SQD_HIP_CHECK(hipGetLastError());
  // rows in several pieces: inside a Davidson run the first reader of the new vector adds the partial rows (opp_split)
  if (s->n_multi > 0 && !(c->sigma_defer_reduce && indexed)) {
    OppReduceArgs r;
    r.rows = s->multi.as<MultiRow>();
    r.partial = s->partial.as<double>();
    r.sigma = d_sigma;
    r.nb = c->nb;
    r.stop = g.stop;
    r.vec_index = g.vec_index;
    r.s_stride = out_stride;
    hipLaunchKernelGGL(k_opp_reduce, dim3((unsigned)s->n_multi, (unsigned)((c->nb + 1023) / 1024)), dim3(256), 0, c->stream, r);
    SQD_HIP_CHECK(hipGetLastError());
  }
  if (c->ev_after_sigma_kernel) {
    SQD_HIP_CHECK(hipEventRecord(c->ev_after_sigma_kernel, c->stream));
    c->ev_after_sigma_kernel = nullptr;
  }
  return SQD_OK;
}

// the split-row records of the latest opp_select (for k_dots_eig's deferred sum); false: every row is in one piece
bool opp_split(const sqd_ctx* c, const int32_t** rowinfo, const double** partial) {
  const OppState* s = static_cast<const OppState*>(c->opp);
  if (!s || !c->sig_opp || s->n_multi == 0) return false;
  *rowinfo = s->rowinfo.as<int32_t>();
  *partial = s->partial.as<double>();
  return true;
}

}  // namespace sqd
