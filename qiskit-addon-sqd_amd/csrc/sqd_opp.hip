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
// Why a kernel of its own.  The work-item kernel (sqd_sigma.hip) spends 24 vector instructions per multiply-add at 3000
// strings per spin (rocprofv3 counters, profiles/r05/pmc_spmm_hf3000_counters.txt: 422 M VALU instructions for 1.1e9
// multiply-adds; waves parked 55 % of their life): every item of <= 3 alpha links re-reads and re-decodes all 33 000 beta
// link records, runs them through virtual rows and LDS partial sums, and writes a partial row that a later launch adds.
// Here ONE workgroup owns a target row A (and a range of columns):
//   * every thread keeps its share of the beta link list -- S <= 12 consecutive links, one packed 32-bit record each
//     {source column, orbital pair, sign, last-of-column} -- in registers for the whole row, and one accumulator per link;
//   * the row's entries are staged four at a time, INTERLEAVED: Cst[B'][4] (signed source rows) and Wst[rs][4] (weight
//     rows), so that a link costs two 16-byte LDS gathers per operand and four multiply-adds -- no address arithmetic
//     per entry, no record traffic, no partial sums per entry batch;
//   * after the last batch the per-link sums are folded to columns in a fixed order (runs inside a thread, then the
//     threads a column spans, oldest first): one sigma row, written once.  No partial rows, no reduce launch.
// Rows of more than 3072 columns (k_opp_rows<RM, true>, RM = 4 .. 8 staged columns per thread, sets of up to 8192 strings):
// the J rows ride in registers for the workgroup's OWN column range only (<= 2 columns per thread) and the alpha single x
// beta occupation term is formed behind the barrier from the staged, signed source values.
// Bound: latency of a one-workgroup-per-CU batch loop (profiles/r05/opp_probe.txt: neither the LDS gathers nor the
// request latency of a batch alone).
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <numeric>

#include "sqd_common.h"

namespace sqd {

constexpr int OPP_K = 2;      // entries staged per batch (the interleave width of Cst / Wst: one 16-byte gather per operand; 4 measured slower: 219 | 817 | 2319 us per sigma at 1000^2 | 2000^2 | 3000^2 against 195 | 763 | 2209)
static_assert(OPP_K % 2 == 0, "entries are staged and gathered in pairs");
constexpr int OPP_SMAX = 12;  // beta links per thread (registers: a packed record + an accumulator each, beside the staged batch)
constexpr int OPP_SMAX3 = 11; // ... with three staged columns per thread (twelve: two spilled registers in the spin-penalty instantiation)
constexpr int OPP_RREG = 3;   // staged columns per thread (nb <= 3 * threads: longer rows run k_opp_src, sqd_oppsrc.hip)
constexpr uint32_t OPP_SIGN = 1u << 28, OPP_LAST = 1u << 29, OPP_LIVE = 1u << 30, OPP_DIR = 1u << 31;  // DIR: low bit of the link's widx

// one workgroup's share of a row: entries [e0, e0 + ne) of row A (entry 0 = the row itself, entry e > 0 = its alpha single
// link e - 1); slot < 0: the row has this one item and is written in place, else partial row `slot` (added in slot
// order by the first reader of the vector -- k_dots_eig inside a Davidson run, k_opp_reduce otherwise)
struct OppItem {
  uint32_t A;
  int32_t e0, ne, slot;
};
struct OppState {
  DevBuf link, lcol, back, items, halves, rowinfo, partial, multi;
  std::vector<OppItem> h_items;
  std::vector<int32_t> h_rowinfo;
  std::vector<MultiRow> h_multi;
  std::vector<int64_t> h_halves;  // [H + 1] first link of every column range, then [H + 1] first column
  int H = 1, S = 0, T = 0;
  int64_t n_items = 0, n_slots = 0, n_multi = 0;
  size_t shmem = 0;
};

void opp_release(sqd_ctx* c) {
  oppsrc_release(c);
  if (!c->opp) return;
  OppState* s = static_cast<OppState*>(c->opp);
  for (DevBuf* b : {&s->link, &s->lcol, &s->back, &s->items, &s->halves, &s->rowinfo, &s->partial, &s->multi}) b->release();
  delete s;
  c->opp = nullptr;
}

// ---- tables: thread t of column range h owns the links l0[h] + t S .. + S - 1 (consecutive: a run of whole columns and
// two partial ones); link[h][s][t] packed record, lcol[h][s][t] its target column, back[h][t] = how many preceding
// threads hold earlier links of the column thread t starts in
struct OppTabArgs {
  GPtr<const int64_t> sb_ptr;
  GPtr<const SRec> sb_rec;
  GPtr<const uint32_t> sb_row;
  GPtr<const int64_t> halves;  // [H + 1]
  GPtr<uint32_t> link, lcol, back;
  int H, S, T;
};
__global__ void __launch_bounds__(256) k_opp_tab(const OppTabArgs g) {
  const int h = blockIdx.y;
  const int64_t l0 = g.halves[h], l1 = g.halves[h + 1];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= g.T) return;
  const int64_t base = ((int64_t)h * g.S) * g.T;
  for (int s = 0; s < g.S; ++s) {
    const int64_t l = l0 + (int64_t)t * g.S + s;
    uint32_t rec = 0u, col = 0u;
    if (l < l1) {
      const SRec r = g.sb_rec[l];
      col = g.sb_row[l];
      rec = (r.src & 0xffffu) | ((srec_widx(r.meta) >> 1) << 16) | ((r.meta >> 31) ? OPP_SIGN : 0u) | OPP_LIVE |
            ((l + 1 == g.sb_ptr[col + 1]) ? OPP_LAST : 0u) | ((srec_widx(r.meta) & 1u) ? OPP_DIR : 0u);
    }
    g.link[base + (int64_t)s * g.T + t] = rec;
    g.lcol[base + (int64_t)s * g.T + t] = col;
  }
  uint32_t bk = 0;
  const int64_t lf = l0 + (int64_t)t * g.S;
  if (lf < l1) {
    const int64_t cstart = g.sb_ptr[g.sb_row[lf]];  // first link of the column thread t starts in (>= l0: ranges are cut at columns)
    bk = (uint32_t)(t - (int)((cstart - l0) / g.S));
  }
  g.back[(int64_t)h * g.T + t] = bk;
}

struct OppArgs {
  GPtr<const double> c;
  GPtr<double> sigma, partial;
  GPtr<const double> hdiag, gdense, ja_row, jbT, eri_pp;
  GPtr<const int64_t> sa_ptr;
  GPtr<const SRec> sa_rec;
  GPtr<const uint32_t> link, lcol, back;
  GPtr<const OppItem> items;
  GPtr<const int64_t> colcut;  // [H + 1] first column of every range
  int64_t na, nb;
  int nnorb, S, T, H;
  unsigned n_items;
  GPtr<const int> stop, vec_index;
  int64_t c_stride, s_stride;
  // the linear spin penalty, sigma = (H + shift (S^2 - ss)) c (pyscf's fix_spin_ form for ss < sz(sz+1) + 0.1; SPIN
  // instantiations): S^2 = sz(sz+1) + sum_p n_pb (1 - n_pa) - sum_{p != q} Ea_qp Eb_pq -- a diagonal term on the own row
  // and -shift on the weight of the ONE beta link with the alpha link's orbital pair and the opposite direction
  double ss, shift, szterm;
  GPtr<const uint64_t> strs_a, strs_b;
};

template <int RM, bool SPIN>
__global__ void __launch_bounds__(1024) k_opp_rows(const OppArgs g) {
  constexpr int SM = RM >= 3 ? OPP_SMAX3 : OPP_SMAX;  // links per thread
  static_assert(RM <= OPP_RREG, "rows of more than 3072 columns run k_opp_src (sqd_oppsrc.hip): the 4-8-column instantiations spilled");
  constexpr int RA = RM;  // accumulators of the alpha single x beta occupation term
  HIP_DYNAMIC_SHARED(double, smem)  // Cst[nb][2] | Wst[nn][2]; after the last batch outb[nb] | tailb[T] take Cst's place
  if (g.stop && *g.stop) return;
  // workgroup b runs on XCD b mod 8: the H column ranges of one item -- they stage the same source rows and J rows --
  // take ids 8 apart, i.e. the same XCD at the same time, so that its L2 serves all but the first of them
  const unsigned xcd = blockIdx.x & 7u, kq = blockIdx.x >> 3;
  const int h = (int)(kq % (unsigned)g.H);
  const unsigned item_index = (kq / (unsigned)g.H) * 8u + xcd;
  if (item_index >= g.n_items) return;
  const int T = g.T, tid = threadIdx.x;
  const OppItem it = g.items[item_index];
  const int64_t A = it.A;
  const int64_t nb = g.nb;
  const int nn = g.nnorb;
  const int64_t vsel = g.vec_index ? (int64_t)(*g.vec_index - 1) : 0;
  const double* __restrict__ C = g.c + vsel * g.c_stride;
  double* __restrict__ sig = g.sigma + vsel * g.s_stride;
  const int64_t nbe = (nb + 1) & ~int64_t(1);
  double* Cst = smem;
  double* Wst = Cst + (nbe > (nb + T + 1) / 2 ? nbe : (nb + T + 2) / 2) * OPP_K;  // (room for outb + tailb in Cst's place)
  double* outb = smem;
  double* tailb = smem + nbe;
  const int64_t B0 = g.colcut[h], B1 = g.colcut[h + 1];  // this workgroup writes the columns [B0, B1)
  // this thread's links: packed records in registers for the whole item
  uint32_t rec[SM];
  const uint32_t* __restrict__ lk = g.link + ((int64_t)h * g.S) * T + tid;
#pragma unroll
  for (int s = 0; s < SM; ++s) rec[s] = (s < g.S) ? lk[(int64_t)s * T] : 0u;
  double acc[SM];
#pragma unroll
  for (int s = 0; s < SM; ++s) acc[s] = 0.0;
  double a3[RA];
#pragma unroll
  for (int r = 0; r < RA; ++r) a3[r] = 0.0;
  const int64_t k0 = g.sa_ptr[A];
  const int e_end = it.e0 + it.ne;
  // Register-staged double buffering: the global loads of batch b + 1 (source rows, J rows, weight rows: everything a
  // thread stages) are requested right behind the barrier that publishes batch b and land while batch b's links are
  // gathered from LDS; a workgroup that fills the CU's registers runs alone on it, so nothing else hides that latency.
  double px[RM][OPP_K], pjb[RM][OPP_K], pw[OPP_K];
  double psg[OPP_K];
  bool plnk[OPP_K];
  uint32_t ppart[OPP_K];  // SPIN: {pair, direction, LIVE} of the beta link an entry's alpha link pairs with in S^2, in the
                          // bit layout of the link records (0: none -- no live record matches it)
  const double* pwrow[OPP_K];
  auto request = [&](int e0) {
    const double* srow[OPP_K];
    const double* jrow[OPP_K];
#pragma unroll
    for (int j = 0; j < OPP_K; ++j) {
      const int e = e0 + j;
      const bool valid = e < e_end;
      plnk[j] = valid && e > 0;
      SRec r = SRec{(uint32_t)A, 0u};
      if (plnk[j]) r = g.sa_rec[k0 + e - 1];
      const int64_t pair = (int64_t)(srec_widx(r.meta) >> 1);
      if constexpr (SPIN)
        ppart[j] = plnk[j] ? (((uint32_t)pair << 16) | ((srec_widx(r.meta) & 1u) ? 0u : OPP_DIR) | OPP_LIVE) : 0u;
      srow[j] = C + (int64_t)r.src * nb;
      psg[j] = valid ? (plnk[j] ? srec_sign(r.meta) : 1.0) : 0.0;
      pwrow[j] = plnk[j] ? g.eri_pp + pair * nn : g.ja_row + A * nn;
      jrow[j] = g.jbT + pair * nb;
    }
    const int iw = tid < nn ? tid : nn - 1;
#pragma unroll
    for (int j = 0; j < OPP_K; ++j) pw[j] = pwrow[j][iw];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int64_t B = tid + (int64_t)r * T;
      const int64_t Bc = B < nb ? B : nb - 1;
#pragma unroll
      for (int j = 0; j < OPP_K; ++j) {
        px[r][j] = srow[j][Bc];
        pjb[r][j] = jrow[j][Bc];
      }
    }
  };
  // registers -> LDS (signed source rows and weight rows, interleaved); the alpha single x beta occupation term rides
  // on the pass over the source rows (own columns)
  auto park = [&]() {
    if (tid < nn) {
#pragma unroll
      for (int j = 0; j < OPP_K; j += 2)
        *reinterpret_cast<double2*>(Wst + (int64_t)tid * OPP_K + j) =
            make_double2(psg[j] != 0.0 ? pw[j] : 0.0, psg[j + 1] != 0.0 ? pw[j + 1] : 0.0);
    }
    for (int i = tid + T; i < nn; i += T) {  // (more orbital pairs than threads: norb > 44 with 1024 threads)
#pragma unroll
      for (int j = 0; j < OPP_K; j += 2)
        *reinterpret_cast<double2*>(Wst + (int64_t)i * OPP_K + j) =
            make_double2(psg[j] != 0.0 ? pwrow[j][i] : 0.0, psg[j + 1] != 0.0 ? pwrow[j + 1][i] : 0.0);
    }
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int64_t B = tid + (int64_t)r * T;
      if (B < nb) {
        double x[OPP_K];
#pragma unroll
        for (int j = 0; j < OPP_K; ++j) {
          x[j] = px[r][j] * psg[j];
          a3[r] += plnk[j] ? pjb[r][j] * x[j] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < OPP_K; j += 2) *reinterpret_cast<double2*>(Cst + B * OPP_K + j) = make_double2(x[j], x[j + 1]);
      }
    }
  };
  request(it.e0);
  for (int e0 = it.e0; e0 < e_end; e0 += OPP_K) {
    park();
    __syncthreads();
    uint32_t cpart[OPP_K];
    if constexpr (SPIN) {
#pragma unroll
      for (int j = 0; j < OPP_K; ++j) cpart[j] = ppart[j];
    }
    if (e0 + OPP_K < e_end) request(e0 + OPP_K);
#pragma unroll
    for (int s = 0; s < SM; ++s) {
      const uint32_t rc = rec[s];
      const double* cp = Cst + (rc & 0xffffu) * OPP_K;
      const double* wp = Wst + ((rc >> 16) & 0xfffu) * OPP_K;
      double2 cv[OPP_K / 2], wv2[OPP_K / 2];
#pragma unroll
      for (int j = 0; j < OPP_K / 2; ++j) {
        cv[j] = *reinterpret_cast<const double2*>(cp + 2 * j);
        wv2[j] = *reinterpret_cast<const double2*>(wp + 2 * j);
      }
      if constexpr (SPIN) {
        constexpr uint32_t KEY = OPP_DIR | OPP_LIVE | 0x0fff0000u;  // (a dead slot, rc = 0, matches "none": its sum is never read)
        const double pen = -g.shift;
#pragma unroll
        for (int j = 0; j < OPP_K / 2; ++j) {
          wv2[j].x += (((rc ^ cpart[2 * j]) & KEY) == 0u) ? pen : 0.0;
          wv2[j].y += (((rc ^ cpart[2 * j + 1]) & KEY) == 0u) ? pen : 0.0;
        }
      }
#pragma unroll
      for (int j = 0; j < OPP_K / 2; ++j) {
        acc[s] += wv2[j].x * cv[j].x;
        acc[s] += wv2[j].y * cv[j].y;
      }
    }
    __syncthreads();
  }
  // ---- per-link sums -> columns, fixed order.  Runs inside the thread; a column that spans threads is finished by the
  // thread that holds its last link, which adds the open runs of the threads before it, oldest first.
  for (int64_t B = tid; B < nb; B += T) outb[B] = 0.0;  // (Cst is done with: the loop's last barrier)
  __syncthreads();
  const uint32_t* __restrict__ lc = g.lcol + ((int64_t)h * g.S) * T + tid;
  const uint32_t bk = g.back[(int64_t)h * T + tid];
  double run = 0.0, firstrun = 0.0;
  int firstcol = -1;
  bool first = true, open = false;
#pragma unroll
  for (int s = 0; s < SM; ++s) {
    const uint32_t rc = rec[s];
    if (rc & OPP_LIVE) {
      run += (rc & OPP_SIGN) ? -acc[s] : acc[s];
      open = true;
      if (rc & OPP_LAST) {
        const uint32_t col = lc[(int64_t)s * T];
        if (first && bk > 0) {
          firstrun = run;
          firstcol = (int)col;
        } else {
          outb[col] = run;
        }
        run = 0.0;
        first = false;
        open = false;
      }
    }
  }
  tailb[tid] = open ? run : 0.0;
  __syncthreads();
  if (firstcol >= 0) {
    double sum = 0.0;
    for (uint32_t j = bk; j >= 1; --j) sum += tailb[tid - (int)j];
    outb[firstcol] = sum + firstrun;
  }
  __syncthreads();
  const bool has0 = it.e0 == 0;  // the piece that holds the row itself also brings the diagonal and the same-spin product
  const double* __restrict__ crow = C + A * nb;
  const double* __restrict__ hd = g.hdiag + A * nb;
  const double* __restrict__ gd = g.gdense + A * nb;
  double* __restrict__ orow = it.slot < 0 ? sig + A * nb : g.partial + (int64_t)it.slot * nb;
#pragma unroll
  for (int r = 0; r < RA; ++r) {
    const int64_t B = tid + (int64_t)r * T;
    if (B >= B0 && B < B1) {
      double v = a3[r] + outb[B];
      if (has0) {
        double d = hd[B];
        if constexpr (SPIN) d += g.shift * (g.szterm + (double)__popcll(g.strs_b[B] & ~g.strs_a[A]) - g.ss);
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
static size_t opp_shmem(int64_t nb, int nn, int T) {
  const int64_t nbe = (nb + 1) & ~int64_t(1);
  const int64_t crows = nbe > (nb + T + 1) / 2 ? nbe : (nb + T + 2) / 2;  // Cst, or outb + tailb in its place
  return (size_t)(crows * OPP_K + (int64_t)((nn + 1) & ~1) * OPP_K) * 8;
}

// k_opp_rows on this subspace?  (rows of <= 3 columns per thread, <= 64 column ranges)
static bool opp_rows_select(sqd_ctx* c, int64_t na, int64_t nb, const int64_t* tot) {
  const int64_t L = tot[2];  // beta single links
  if (L < 1 || nb > 65535 || c->nnorb > 4095) return false;
  if (!c->opp) c->opp = new OppState();
  OppState* s = static_cast<OppState*>(c->opp);
  // threads: 512 for short rows (four workgroups per CU), 1024 beyond (two); column ranges H so that a thread holds at
  // most OPP_SMAX links
  int T = (nb <= (int64_t)OPP_RREG * 512 && L <= (int64_t)OPP_SMAX * 512) ? 512 : 1024;
  if (const char* env = std::getenv("SQD_OPP_T")) {  // tuning / test hook (small workgroups: several column ranges on small sets)
    const int v = std::atoi(env);
    if (v >= 64 && v <= 1024 && v % 64 == 0) T = v;
  }
  int smax = OPP_SMAX;
  if (const char* env = std::getenv("SQD_OPP_S")) {  // test hook: links per thread
    const int v = std::atoi(env);
    if (v >= 1 && v <= OPP_SMAX) smax = v;
  }
  if (nb > (int64_t)OPP_RREG * T) return false;  // (longer rows: k_opp_src)
  if ((nb + T - 1) / T >= 3 && smax > OPP_SMAX3) smax = OPP_SMAX3;
  if (opp_shmem(nb, c->nnorb, T) + 1024 > (size_t)c->lds_bytes) return false;
  // cut the link list at column starts, as evenly as the columns allow; one range more while the longest does not fit
  const int64_t* ps = c->h_sptr_b;
  int H = (int)((L + (int64_t)smax * T - 1) / ((int64_t)smax * T)), S = 0;
  for (;; ++H) {
    if (H > 64) return false;
    s->h_halves.assign((size_t)2 * (H + 1), 0);
    int64_t B = 0, longest = 0;
    for (int h = 1; h <= H; ++h) {
      const int64_t want = (h == H) ? L : (L * h) / H;
      while (B < nb && ps[B] < want) ++B;
      if (h == H) B = nb;
      s->h_halves[h] = ps[B];
      s->h_halves[(size_t)(H + 1) + h] = B;
      longest = std::max(longest, s->h_halves[h] - s->h_halves[h - 1]);
    }
    S = (int)((longest + T - 1) / T);
    if (S <= smax) break;
  }
  s->H = H;
  s->S = S < 1 ? 1 : S;
  s->T = T;
  s->shmem = opp_shmem(nb, c->nnorb, T);
  // work items: a row's entries (itself + its alpha single links) in pieces of at most E, so that the rows of the
  // Hartree-Fock neighbourhood (up to 177 entries) do not run as one workgroup's chain of 90 batches; a row in one
  // piece is written in place, the others as partial rows added in slot order.  Longest pieces first.
  // (measured, HF-centred N x N, us per sigma for E = 4 | 8 | 16 | 32: 1000: 260 | 222 | 202 | 195; 2000: 940 | 828 | 779 |
  // 763; 3000: 2747 | 2437 | 2278 | 2209 -- profiles/r05/opp_probe.txt)
  int E = 32;
  if (const char* env = std::getenv("SQD_OPP_E")) {  // tuning hook
    const int v = std::atoi(env);
    if (v >= OPP_K && v <= 4096) E = v / OPP_K * OPP_K;
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
  return true;
}

// phase 2 of set_subspace, behind spmm_select: is the opposite-spin part of this subspace taken by whole rows -- k_opp_rows
// (rows of <= 3072 columns) or k_opp_src (sqd_oppsrc.hip: longer rows; SQD_OPP_SRC=1 forces it, =0 forbids it)?
// (SQD_SIGMA_OPP=0 forbids both: the work items then add G as they add the matrix-core product.)
bool opp_select(sqd_ctx* c, int64_t na, int64_t nb, const int64_t* tot) {
  c->sig_opp = false;
  c->opp_src = false;
  if (!c->sig_spmm) return false;
  if (const char* env = std::getenv("SQD_SIGMA_OPP"))
    if (std::atoi(env) == 0) return false;
  int want_src = -1;
  if (const char* env = std::getenv("SQD_OPP_SRC")) want_src = std::atoi(env) != 0 ? 1 : 0;
  if (want_src != 1 && opp_rows_select(c, na, nb, tot)) {
    c->sig_opp = true;
    return true;
  }
  if (want_src != 0 && oppsrc_select(c, na, nb, tot)) {
    c->sig_opp = c->opp_src = true;
    return true;
  }
  return false;
}

int opp_build(sqd_ctx* c) {
  if (c->opp_src) return oppsrc_build(c);
  OppState* s = static_cast<OppState*>(c->opp);
  const SpinTables& tb = c->sp[1];
  const size_t nrec = (size_t)s->H * s->S * s->T;
  SQD_TRY(s->link.reserve(nrec * 4 + 64));
  SQD_TRY(s->lcol.reserve(nrec * 4 + 64));
  SQD_TRY(s->back.reserve((size_t)s->H * s->T * 4 + 64));
  SQD_TRY(s->items.reserve((size_t)s->n_items * sizeof(OppItem) + 64));
  SQD_TRY(s->rowinfo.reserve((size_t)2 * c->na * 4 + 64));
  SQD_TRY(s->multi.reserve((size_t)s->n_multi * sizeof(MultiRow) + 64));
  SQD_TRY(s->partial.reserve((size_t)s->n_slots * c->nb * 8 + 64));
  SQD_TRY(s->halves.reserve(s->h_halves.size() * 8));
  SQD_HIP_CHECK(hipMemcpyAsync(s->items.p, s->h_items.data(), (size_t)s->n_items * sizeof(OppItem), hipMemcpyHostToDevice, c->stream));
  SQD_HIP_CHECK(hipMemcpyAsync(s->rowinfo.p, s->h_rowinfo.data(), (size_t)2 * c->na * 4, hipMemcpyHostToDevice, c->stream));
  if (s->n_multi)
    SQD_HIP_CHECK(hipMemcpyAsync(s->multi.p, s->h_multi.data(), (size_t)s->n_multi * sizeof(MultiRow), hipMemcpyHostToDevice, c->stream));
  SQD_HIP_CHECK(hipMemcpyAsync(s->halves.p, s->h_halves.data(), s->h_halves.size() * 8, hipMemcpyHostToDevice, c->stream));
  OppTabArgs a;
  a.sb_ptr = tb.s_ptr.as<int64_t>();
  a.sb_rec = tb.s_rec.as<SRec>();
  a.sb_row = tb.s_row.as<uint32_t>();
  a.halves = s->halves.as<int64_t>();
  a.link = s->link.as<uint32_t>();
  a.lcol = s->lcol.as<uint32_t>();
  a.back = s->back.as<uint32_t>();
  a.H = s->H;
  a.S = s->S;
  a.T = s->T;
  hipLaunchKernelGGL(k_opp_tab, dim3((unsigned)((s->T + 255) / 256), (unsigned)s->H), dim3(256), 0, c->stream, a);
  SQD_HIP_CHECK(hipGetLastError());
  return SQD_OK;
}

// sigma = (hdiag + opposite-spin part) c + G, G = sqd_ctx::gdense as spmm_launch has just formed it for the same vector
int opp_launch(sqd_ctx* c, const double* d_c, double* d_sigma, int64_t in_stride, int64_t out_stride, bool spin, double ss,
               double shift) {
  if (c->opp_src) return oppsrc_launch(c, d_c, d_sigma, in_stride, out_stride, spin, ss, shift);
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
  g.link = s->link.as<uint32_t>();
  g.lcol = s->lcol.as<uint32_t>();
  g.back = s->back.as<uint32_t>();
  g.items = s->items.as<OppItem>();
  g.partial = s->partial.as<double>();
  g.colcut = s->halves.as<int64_t>() + (s->H + 1);
  g.na = c->na;
  g.nb = c->nb;
  g.nnorb = c->nnorb;
  g.S = s->S;
  g.T = s->T;
  g.H = s->H;
  g.stop = c->sigma_stop;
  const bool indexed = c->sigma_index && (in_stride || out_stride);
  g.vec_index = indexed ? c->sigma_index : nullptr;
  g.c_stride = in_stride;
  g.s_stride = out_stride;
  g.ss = ss;
  g.shift = shift;
  {
    const double sz = 0.5 * (c->nelec[0] - c->nelec[1]);
    g.szterm = sz * (sz + 1.0);
  }
  g.strs_a = c->sp[0].strs.as<uint64_t>();
  g.strs_b = c->sp[1].strs.as<uint64_t>();
  if (s->shmem > 64 * 1024) {
    static std::atomic<size_t> granted[64];
    const int dev = c->device & 63;
    if (s->shmem > granted[dev].load(std::memory_order_relaxed)) {
#define SQD_OPP_F(RM_) reinterpret_cast<const void*>(&k_opp_rows<RM_, false>), reinterpret_cast<const void*>(&k_opp_rows<RM_, true>)
      for (const void* f : {SQD_OPP_F(1), SQD_OPP_F(2), SQD_OPP_F(3)})
#undef SQD_OPP_F
        SQD_HIP_CHECK(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->shmem));
      granted[dev].store(s->shmem, std::memory_order_relaxed);
    }
  }
  const int rm = (int)((c->nb + s->T - 1) / s->T);  // columns per thread in the coalesced passes (<= OPP_RREG: opp_rows_select)
  g.n_items = (unsigned)s->n_items;
  const dim3 grid(8u * (unsigned)((s->n_items + 7) / 8) * (unsigned)s->H), block((unsigned)s->T);
#define SQD_OPP_GO(RM_)                                                                            \
  do {                                                                                             \
    if (spin) hipLaunchKernelGGL((k_opp_rows<RM_, true>), grid, block, s->shmem, c->stream, g);    \
    else hipLaunchKernelGGL((k_opp_rows<RM_, false>), grid, block, s->shmem, c->stream, g);        \
  } while (0)
  if (rm <= 1) SQD_OPP_GO(1);
  else if (rm == 2) SQD_OPP_GO(2);
  else SQD_OPP_GO(3);
#undef SQD_OPP_GO
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
  if (c->opp_src) return c->sig_opp && oppsrc_split(c, rowinfo, partial);
  const OppState* s = static_cast<const OppState*>(c->opp);
  if (!s || !c->sig_opp || s->n_multi == 0) return false;
  *rowinfo = s->rowinfo.as<int32_t>();
  *partial = s->partial.as<double>();
  return true;
}

}  // namespace sqd
