// sigma = P H P c on the alpha x beta product subspace (the Davidson matvec), plus P S^2 P c.
//
// Replaces pyscf selected_ci.contract_2e (SCIcontract_2e_aaaa x2 + SCIcontract_2e_bbaa, with its
// per-call O(norb^4) integral re-packing and two transposes of C) and selected_ci.contract_ss /
// the fix_spin_ penalty; reference call sites qiskit_addon_sqd/fermion.py:721-723, :810-818, :830.
//
// Formulation (exact, no dense (string x norb^2) intermediate, no multiplications by zero):
//   sigma[A,B] = hdiag[A,B] C[A,B]
//     + sum_{A'}  Ha[A,A'] C[A',B]                                  same-spin alpha (singles+doubles)
//     + sum_{B'}  Hb[B,B'] C[A,B']                                  same-spin beta
//     + sum_{(A',pq,s) in Sa(A)} s * Jb[B][pq] * C[A',B]            alpha single x beta occupation
//     + sum_{(B',rs,t) in Sb(B)} t * Ja[A][rs] * C[A,B']            beta single x alpha occupation
//     + sum_{Sa(A)} sum_{Sb(B)} s t (pq|rs) C[A',B']                single x single
//   (S^2 adds  -sum Ea_qp Eb_pq  = one extra entry in the (pq|..) row, and a diagonal term.)
//
// gfx950 mapping: the launch is a list of work items (built once per subspace, sqd_tables.hip), one
// workgroup each, so that the few highly connected strings around the Hartree-Fock determinant do not
// serialise it.  A workgroup keeps its slice of one sigma row in registers.  For a batch of up to K
// alpha single links it stages, coalesced, the K source rows C[A',:] and the K integral rows (pq|:)
// into LDS, then every lane walks the sliced-ELL single-excitation list of its beta string and gathers
// from LDS.  Global memory is only ever read with unit stride; all irregular accesses hit LDS.
#include <atomic>
#include <cmath>
#include <cstdlib>

#include "sqd_common.h"
#include "sqd_direct.h"

namespace sqd {

struct SigmaArgs {
  GPtr<const double> c;
  GPtr<double> sigma;
  GPtr<double> partial;
  GPtr<const WorkItem> items;
  int64_t na, nb;
  int64_t row0;  // first alpha row of this context's shard: hdiag and sigma are indexed relative to it
  int nnorb, nb_pad, K;
  int mode;  // 0: H (+ penalty when spin), 1: pure S^2
  int type_mask;  // profiling hook (env SQD_SIGMA_TYPES): bit t set = execute work items of type t; default 7
  int spin;
  double ss, shift, szterm;
  GPtr<const uint64_t> strs_a;
  GPtr<const uint64_t> strs_b;
  GPtr<const double> hdiag;
  GPtr<const SRec> sa_rec;  // alpha singles (CSR order)
  GPtr<const uint32_t> ha_src;  // alpha merged same-spin links
  GPtr<const double> ha_val;
  GPtr<const double> ja_row;
  // beta lists as capped sliced ELL over virtual rows (sqd_tables.hip); own[3B..] = {first full row,
  // number of full rows, tail row or -1} of string B
  GPtr<const int32_t> vs_cnt;
  GPtr<const int32_t> vs_own;
  GPtr<const int32_t> vd_cnt;
  GPtr<const int32_t> vd_own;
  int nv_s, nv_d;
  // column chunks (gridDim.y): chunk k covers columns [k*chunk_cols, ...) and the virtual rows
  // [v*_chunk[k], v*_chunk[k+1]); nvs_max / nvd_max = capacity of the LDS partial-sum arrays: the longest
  // such range, or less -- then the range is walked in passes of that many virtual rows
  GPtr<const int32_t> vs_chunk;
  GPtr<const int32_t> vd_chunk;
  int64_t chunk_cols;
  int nvs_max, nvd_max;
  GPtr<const int64_t> esb_sl;
  GPtr<const SRec> esb_rec;
  GPtr<const double> esb_val;
  GPtr<const int64_t> edb_sl;
  GPtr<const uint32_t> edb_src;
  GPtr<const double> edb_val;
  GPtr<const double> jbT;
  GPtr<const double> eri_pp;
  // Davidson enqueues whole iterations ahead of the host; when the device finds the solve finished it raises this
  // flag and the launch returns at once (nullptr: unconditional)
  GPtr<const int> stop;
  // device-controlled Davidson: the vector to work on is chosen on the device.  With vec_index != nullptr the
  // input is c + (*vec_index - 1) * c_stride and the output sigma + (*vec_index - 1) * s_stride.
  GPtr<const int> vec_index;
  int64_t c_stride, s_stride;
  // dense same-spin mode (sqd_ctx::sig_dense): the matrix-core product H_a C + C H_b of this vector, row-major like
  // sigma, as DENSE_SPLIT partial products over disjoint k ranges (gdense[s * gdense_stride + ...]); the own-row items
  // add them in order s = 0, 1, ... and carry no same-spin link of their own.  nullptr: sparse same-spin links.
  GPtr<const double> gdense;
  int64_t gdense_stride;
  int gsplit;  // partial products to add: DENSE_SPLIT (matrix cores) or 1 (the sparse product of sqd_spmm.hip)
  // row-sharded solves, sigma in two launches around the all-gather of the input vector (sqd_shard_dav_sigma_part):
  //   own_mode 1 -- in front of the gather: own-row items only, on the rows this rank owns (c_own: the rank's own rows of
  //                 the vector, (row0 .. row1) x nb), without their folded alpha links;
  //   own_mode 2 -- behind it: every other item, and the own-row items add their folded alpha links onto the element
  //                 they have written (a_partial + chunk sum: the order of the one-launch kernel, the same bits).
  // mask_skip: items outside type_mask return without writing (the profiling hook writes zeros instead).
  GPtr<const double> c_own;
  int own_mode, mask_skip;
  // launch geometry of THIS subspace: threads that work (a batched launch uses the largest workgroup of its class; the
  // surplus threads of a smaller subspace idle), work items, column chunks
  int T;
  unsigned gx, gy;
};

// N consecutive same-spin links starting at l0: a += val[l] * C[src[l], B] in link order.  N is a compile-time
// constant and nothing is predicated: the N (wave-uniform, scalar) record loads are issued together, then the
// N row loads, then the multiply-adds.  With a test per link the compiler wrapped every link in its own block
// {s_load src; wait; global_load; s_load val; wait} -- two scalar round trips per link in sequence (ISA).
template <int N>
__device__ inline double axpy_round(const double* __restrict__ C, const uint32_t* __restrict__ src,
                                    const double* __restrict__ val, int64_t l0, int64_t nb, int64_t B, double a) {
  double x[N], v[N];
  uint32_t s[N];
#pragma unroll
  for (int u = 0; u < N; ++u) {
    s[u] = src[l0 + u];
    v[u] = val[l0 + u];
  }
#pragma unroll
  for (int u = 0; u < N; ++u) x[u] = C[(int64_t)s[u] * nb + B];
#pragma unroll
  for (int u = 0; u < N; ++u) a += v[u] * x[u];
  return a;
}
// sum_l val[l] * C[src[l], B] over a chunk of same-spin links: whole rounds of 8, then the remainder as one
// round of its exact size (no padded loads: at 10^4 x 10^4 the row reads are the bandwidth)
__device__ inline double axpy_chunk(const double* __restrict__ C, const uint32_t* __restrict__ src,
                                    const double* __restrict__ val, int64_t begin, int count, int64_t nb, int64_t B) {
  double a = 0.0;
  int64_t l = begin;
  for (int k = 0; k + 8 <= count; k += 8, l += 8) a = axpy_round<8>(C, src, val, l, nb, B, a);
  switch (count & 7) {
    case 1: a = axpy_round<1>(C, src, val, l, nb, B, a); break;
    case 2: a = axpy_round<2>(C, src, val, l, nb, B, a); break;
    case 3: a = axpy_round<3>(C, src, val, l, nb, B, a); break;
    case 4: a = axpy_round<4>(C, src, val, l, nb, B, a); break;
    case 5: a = axpy_round<5>(C, src, val, l, nb, B, a); break;
    case 6: a = axpy_round<6>(C, src, val, l, nb, B, a); break;
    case 7: a = axpy_round<7>(C, src, val, l, nb, B, a); break;
    default: break;
  }
  return a;
}

// ---- one virtual row of a beta list against LDS-staged data.  Link records are fetched PF at a time
// (independent coalesced loads in flight together), then consumed against LDS.  The LDS phase is written
// WITHOUT per-link predicates: links beyond the row's count get index 0 and weight 0, so that all PF gathers
// of a round are issued back to back and waited for once (predicated per-link blocks make the compiler wait
// for each pair of LDS reads separately -- measured as the pace of the whole kernel).
// singles on the own row: sum (value + sign * W[pair]) * Crow[src]
// (same_spin == false -- dense same-spin mode: the links' own values are in the matrix-core product, only the
// alpha-occupation term sign * W[pair] is left)
__device__ inline double vrow_singles_own(const SigmaArgs& g, int64_t v, const double* Crow, const double* W2,
                                          bool same_spin) {
  const int64_t base = g.esb_sl[v >> 6] + (v & 63);
  const int cnt = g.vs_cnt[v];
  constexpr int PF = 8;
  double a = 0.0;
  for (int k0 = 0; k0 < cnt; k0 += PF) {
    SRec recs[PF];
    double vals[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      recs[u] = SRec{0u, 0u};
      vals[u] = 0.0;
      if (k0 + u < cnt) {
        recs[u] = g.esb_rec[base + (int64_t)(k0 + u) * 64];
        if (same_spin) vals[u] = g.esb_val[base + (int64_t)(k0 + u) * 64];
      }
    }
    double w[PF], x[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      w[u] = W2[srec_widx(recs[u].meta) >> 1];
      x[u] = Crow[recs[u].src];
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const double sg = (k0 + u < cnt) ? srec_sign(recs[u].meta) : 0.0;
      a += (vals[u] + sg * w[u]) * x[u];
    }
  }
  return a;
}
// doubles on the own row: sum value * Crow[src]
__device__ inline double vrow_doubles_own(const SigmaArgs& g, int64_t v, const double* Crow) {
  const int64_t base = g.edb_sl[v >> 6] + (v & 63);
  const int cnt = g.vd_cnt[v];
  constexpr int PF = 8;
  double a = 0.0;
  for (int k0 = 0; k0 < cnt; k0 += PF) {
    uint32_t srcs[PF];
    double vals[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      srcs[u] = 0u;
      vals[u] = 0.0;
      if (k0 + u < cnt) {
        srcs[u] = g.edb_src[base + (int64_t)(k0 + u) * 64];
        vals[u] = g.edb_val[base + (int64_t)(k0 + u) * 64];
      }
    }
    double x[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) x[u] = Crow[srcs[u]];
#pragma unroll
    for (int u = 0; u < PF; ++u) a += vals[u] * x[u];
  }
  return a;
}
// string B's share of a list: its contiguous run of full rows, then its tail (fixed order)
// (part[] holds the rows of one column chunk, whose first row is v0)
// part[] holds the virtual rows [v0, v1) of this pass (all rows of the column chunk when one pass suffices;
// full rows precede tails in the row order, so passes add a string's rows in the same fixed order)
__device__ inline double own_rows_sum(const int32_t* __restrict__ own, int64_t B, const double* part, int v0, int v1) {
  const int f0 = own[3 * B], nfull = own[3 * B + 1], tail = own[3 * B + 2];
  const int x0 = f0 > v0 ? f0 : v0, x1 = (f0 + nfull < v1) ? f0 + nfull : v1;
  double a = 0.0;
  for (int x = x0; x < x1; ++x) a += part[x - v0];
  if (tail >= v0 && tail < v1) a += part[tail - v0];
  return a;
}

// passes needed to walk ns singles' and nd doubles' virtual rows with LDS room for cs / cd partial sums
__device__ inline int npasses(int ns, int cs, int nd, int cd) {
  const int a = (ns + cs - 1) / (cs > 0 ? cs : 1), b = (nd + cd - 1) / (cd > 0 ? cd : 1);
  const int n = a > b ? a : b;
  return n > 1 ? n : 1;
}

// singles against a batch of KT staged alpha links: sum sign * sum_j W[j][pair] * Crow[j][src].
// KT is the batch capacity (compile-time: the j loop unrolls, all gathers of a link are in flight together);
// slots beyond the batch's actual count hold zero rows (staged by the caller).
// SPIN: the S^2 operator couples alpha link j (cre a, des b) to the one beta link with the same orbital
// pair and the opposite direction (cre b, des a); penw[j] is that beta link's widx (-1 for an empty slot),
// pen the coefficient.
// One virtual row's header and first round of records, requested EARLY: a work item's chain of dependent loads is
// item -> alpha records -> rows -> (barrier) -> virtual-row header -> beta records -> ...; the last two depend on
// nothing before them, so they are put in flight beside the first two and are in registers when the staging is done
// (six round trips per workgroup become four -- the kernel lives on resident workgroups x their latency).
struct VRowPre {
  int cnt;
  int64_t base;
  SRec recs[8];
};
__device__ inline void vrow_header(const SigmaArgs& g, int64_t v, bool on, VRowPre& p) {
  p.cnt = on ? g.vs_cnt[v] : 0;
  p.base = on ? g.esb_sl[v >> 6] + (v & 63) : 0;
}
__device__ inline void vrow_records(const SigmaArgs& g, VRowPre& p) {
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    p.recs[u] = SRec{0u, 0u};
    if (u < p.cnt) p.recs[u] = g.esb_rec[p.base + (int64_t)u * 64];
  }
}
template <bool SPIN, int KT>
__device__ inline double vrow_singles_batch(const SigmaArgs& g, int64_t v, const double* Crow, const double* W, int ws,
                                            const int* penw, double pen, const VRowPre& pre, bool use_pre) {
  const int64_t base = use_pre ? pre.base : g.esb_sl[v >> 6] + (v & 63);
  const int cnt = use_pre ? pre.cnt : g.vs_cnt[v];
  constexpr int PF = 8;
  int pw[KT];
#pragma unroll
  for (int j = 0; j < KT; ++j) pw[j] = SPIN ? penw[j] : -1;
  double a = 0.0;
  for (int k0 = 0; k0 < cnt; k0 += PF) {
    SRec recs[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (use_pre && k0 == 0) {
        recs[u] = pre.recs[u];
      } else {
        recs[u] = SRec{0u, 0u};
        if (k0 + u < cnt) recs[u] = g.esb_rec[base + (int64_t)(k0 + u) * 64];
      }
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int widx = (int)srec_widx(recs[u].meta);
      const double* cr = Crow + recs[u].src;
      const double* w = W + (widx >> 1);
      double wj[KT], cj[KT];
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        wj[j] = w[(int64_t)j * ws];
        cj[j] = cr[(int64_t)j * g.nb_pad];
      }
      double t = 0.0;
#pragma unroll
      for (int j = 0; j < KT; ++j) {
        double wv = wj[j];
        if (SPIN) wv += (widx == pw[j]) ? pen : 0.0;
        t += wv * cj[j];
      }
      const double sg = (k0 + u < cnt) ? srec_sign(recs[u].meta) : 0.0;
      a += sg * t;
    }
  }
  return a;
}
// dispatch on the batch capacity of this launch (uniform)
template <bool SPIN>
__device__ inline double vrow_singles_batch_k(const SigmaArgs& g, int K, int64_t v, const double* Crow, const double* W,
                                              int ws, const int* penw, double pen, const VRowPre& pre, bool use_pre) {
  switch (K) {
    case 1: return vrow_singles_batch<SPIN, 1>(g, v, Crow, W, ws, penw, pen, pre, use_pre);
    case 2: return vrow_singles_batch<SPIN, 2>(g, v, Crow, W, ws, penw, pen, pre, use_pre);
    case 3: return vrow_singles_batch<SPIN, 3>(g, v, Crow, W, ws, penw, pen, pre, use_pre);
    default: return vrow_singles_batch<SPIN, 4>(g, v, Crow, W, ws, penw, pen, pre, use_pre);
  }
}

// ---- phase clocks (probe builds only: -DSQD_PHASE_CLOCK; profiles/probes/_sigma_clock.py): thread 0 of every workgroup
// adds the 100 MHz wall-clock deltas of its item's phases into ITS OWN row of a device array (plain stores: atomics on
// shared counters serialised the 1300 workgroups of a launch and were most of what they measured); the host sums the rows
#ifdef SQD_PHASE_CLOCK
constexpr int SCLK_ROWS = 65536, SCLK_COLS = 16;
__device__ unsigned long long sqd_clk_sigma[SCLK_ROWS * SCLK_COLS];
__device__ inline unsigned sclk_row() {
  return ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % SCLK_ROWS;
}
#define SCLK(var) __builtin_amdgcn_s_waitcnt(0); const unsigned long long var = wall_clock64()
#define SCLK_ADD(slot, a, b) do { if (threadIdx.x == 0) sqd_clk_sigma[sclk_row() * SCLK_COLS + ((slot) % 10) + ((slot) >= 10 ? 8 : 0)] += (unsigned long long)((b) - (a)); } while (0)
#else
#define SCLK(var)
#define SCLK_ADD(slot, a, b)
#endif
// LDSROW: the C rows of an item are staged in LDS (the tuned path).  !LDSROW: rows too long for LDS are
// read in place (global memory / L2), one alpha link per batch; everything else is unchanged.
// PASS: the beta lists outgrow the LDS partial-sum arrays and continue in extra passes (staged rows of ~10^4
// strings); a separate instantiation so that the tuned single-pass kernels keep their register budget.
// (bx, by) = work item and column chunk of this workgroup within ITS subspace; T threads work, the others (a batched
// launch is sized for the largest workgroup of its class) park on an index no loop reaches -- they still meet every
// barrier
// ALWAYS inlined into its two kernels (round 6).  As an out-of-line function -- what the compiler chose for the R = 8 / 16
// single-pass instantiations, whose bodies exceed its inline threshold -- the argument record travels through scratch, every
// field is re-loaded per lane with flat loads, the compiler has to treat the workgroup-uniform branches (item type, stop
// flag) as divergent, and the body needs a 400-byte stack frame per thread: the form in which k_sigma<16, ., true, false>
// never returned on the MI355X (profiles/r06/hang_root_cause.txt; tests/test_codeobj_audit.py pins "no kernel of the
// library calls out of line or owns a stack frame").  -DSQD_SIGMA_BODY_NOINLINE rebuilds the old form for the probe.
#ifdef SQD_SIGMA_BODY_NOINLINE
#define SQD_SIGMA_BODY_INLINE __attribute__((noinline))
#else
#define SQD_SIGMA_BODY_INLINE __attribute__((always_inline))
#endif
template <int R, bool SPIN, bool LDSROW, bool PASS>
__device__ SQD_SIGMA_BODY_INLINE inline void sigma_body(const SigmaArgs& g, double* smem, unsigned bx, unsigned by) {
  if (g.stop && *g.stop) return;  // uniform over the subspace
  SCLK(k0);
  const int T = g.T, tid = ((int)threadIdx.x < T) ? (int)threadIdx.x : (1 << 30);
  const WorkItem it = g.items[bx];
  const int64_t A = it.A;
  const int64_t nb = g.nb;
  const int nnorb = g.nnorb;
  // this workgroup's column chunk and the virtual rows owned by its strings
  const int chunk = by;
  const int64_t B0 = (int64_t)chunk * g.chunk_cols;
  const int64_t Bend = (B0 + g.chunk_cols < nb) ? B0 + g.chunk_cols : nb;
  const int vs0 = g.vs_chunk[chunk], vs1 = g.vs_chunk[chunk + 1];
  const int vd0 = g.vd_chunk[chunk], vd1 = g.vd_chunk[chunk + 1];
  // LDS plan (sized in build_sigma_work): [part_s | penw | region]; the region is K x (C row) then
  // K x (integral row) for a batch, or ONE C row, ONE integral row and the doubles' partial sums for an
  // own-row item -- the two uses overlap, so the allocation is their maximum, not their sum
  const int w2s = (nnorb + 1) & ~1;            // one integral row per staged link
  const int nvs_pad = (g.nvs_max + 1) & ~1;
  double* part_s = smem;                                    // [nvs_max] partial sums of the singles' virtual rows
  int* penw = reinterpret_cast<int*>(smem + nvs_pad);       // [K <= 4] S^2 partner widx of each staged link
  double* Crow = smem + nvs_pad + 2;                        // [K][nb_pad]   (LDSROW only)
  const int64_t rows_staged = (it.type == 0) ? 1 : g.K;
  double* W2 = Crow + (LDSROW ? rows_staged * g.nb_pad : 0);  // [rows_staged][w2s]
  double* part_d = W2 + w2s;                                // [nvd_max] (own-row items) ... of the doubles' virtual rows
  const int64_t vsel = g.vec_index ? (int64_t)(*g.vec_index - 1) : 0;
  const double* __restrict__ C = g.c + vsel * g.c_stride;
  double* __restrict__ sigma_out = g.sigma + vsel * g.s_stride;
  // this thread's first virtual row of the beta singles: header now, records as soon as the header is there
  VRowPre pre;
  // (not conditional on the item's type: that would put the header one round trip behind the item record)
  const bool pre_on = LDSROW && vs0 + tid < ((vs0 + g.nvs_max < vs1) ? vs0 + g.nvs_max : vs1);
  vrow_header(g, vs0 + (pre_on ? tid : 0), pre_on, pre);
  double acc[R];
#pragma unroll
  for (int r = 0; r < R; ++r) acc[r] = 0.0;
#ifdef SQD_PHASE_CLOCK
  unsigned long long sclk_mid = 0;
#endif

  if (!((g.type_mask >> it.type) & 1)) {
    if (g.mask_skip) return;  // (uniform over the workgroup; nothing staged, no barrier met yet)
    // profiling hook only: skipped item classes write zeros
  } else if (it.type == 0 && g.own_mode == 2) {
    // behind the all-gather: the folded alpha links of an own-row item, added onto what the first launch wrote
    if (it.count == 0) return;
    double* __restrict__ o2 = (it.slot < 0) ? (sigma_out + (A - g.row0) * nb) : (g.partial + (int64_t)it.slot * nb);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t B = B0 + tid + (int64_t)r * T;
      if (B < Bend) o2[B] += axpy_chunk(C, g.ha_src, g.ha_val, it.begin, it.count, nb, B);
    }
    return;
  } else if (it.type == 0) {
    // ---- own row: slot 0 <- C[A,:], W2 slot 0 <- Ja[A][:]
    const uint64_t sA = g.strs_a[A];
    // the own row and the diagonal are touched exactly once per sigma: stream them past the L2
    // (non-temporal) so that the link lists, which every workgroup re-reads, stay resident
    const double* crow0 = g.c_own ? (const double*)g.c_own + (A - g.row0) * nb : C + A * nb;
    if (LDSROW) {
      for (int64_t i = tid; i < nb; i += T) Crow[i] = __builtin_nontemporal_load(&crow0[i]);
    }
    if (g.mode == 0)
      for (int i = tid; i < nnorb; i += T) {
        W2[i] = g.ja_row[A * nnorb + i];
      }
    __syncthreads();
    SCLK(k1);
    SCLK_ADD(1, k0, k1);  // type 0: item record, own row + J row staged
    // every virtual row of the beta lists, by whichever thread comes next; partial sums through LDS.
    // One pass unless the lists outgrow the partial-sum arrays (long rows, see build_subspace).
    // (the first nvs_max / nvd_max of them here; lists that outgrow the partial-sum arrays -- long rows, see
    // build_subspace -- continue in the extra passes at the end of the kernel)
    const int s1 = (vs0 + g.nvs_max < vs1) ? vs0 + g.nvs_max : vs1;
    const int d1 = (vd0 + g.nvd_max < vd1) ? vd0 + g.nvd_max : vd1;
    if (g.mode == 0) {
      const bool ssl = (g.gdense == nullptr);
      if (LDSROW) {
        for (int v = vs0 + tid; v < s1; v += T) part_s[v - vs0] = vrow_singles_own(g, v, Crow, W2, ssl);
        for (int v = vd0 + tid; v < d1; v += T) part_d[v - vd0] = vrow_doubles_own(g, v, Crow);
      } else {
        for (int v = vs0 + tid; v < s1; v += T) part_s[v - vs0] = vrow_singles_own(g, v, crow0, W2, ssl);
        for (int v = vd0 + tid; v < d1; v += T) part_d[v - vd0] = vrow_doubles_own(g, v, crow0);
      }
    }
    __syncthreads();
#ifdef SQD_PHASE_CLOCK
    SCLK(k2);
    SCLK_ADD(2, k1, k2);  // type 0: virtual rows
    if (threadIdx.x == 0) sqd_clk_sigma[sclk_row() * SCLK_COLS + 0] += 1ull;
    sclk_mid = k2;
#endif
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t B = B0 + tid + (int64_t)r * T;
      if (B < Bend) {
        double d;
        if (g.mode == 0) {
          d = __builtin_nontemporal_load(&g.hdiag[(A - g.row0) * nb + B]);
          if (SPIN) d += g.shift * (g.szterm + (double)__popcll(g.strs_b[B] & ~sA) - g.ss);
        } else {
          d = g.szterm + (double)__popcll(g.strs_b[B] & ~sA);
        }
        double a = d * (LDSROW ? Crow[B] : crow0[B]);
        if (g.mode == 0) {
          // beta same-spin singles (value) + beta single x alpha occupation (W2 slot 0), beta doubles
          a += own_rows_sum(g.vs_own, B, part_s, vs0, s1);
          a += own_rows_sum(g.vd_own, B, part_d, vd0, d1);
          // first same-spin alpha links of this row: unit-stride row reads
          if (g.own_mode != 1) a += axpy_chunk(C, g.ha_src, g.ha_val, it.begin, it.count, nb, B);
          // dense same-spin mode: the whole same-spin part of this element, from the matrix-core product
          if (g.gdense) {
            if (g.gsplit == 1) {  // (uniform)
              a += g.gdense[(A - g.row0) * nb + B];
            } else {
              double gp[DENSE_SPLIT];
#pragma unroll
              for (int sp = 0; sp < DENSE_SPLIT; ++sp) gp[sp] = g.gdense[sp * g.gdense_stride + (A - g.row0) * nb + B];
#pragma unroll
              for (int sp = 0; sp < DENSE_SPLIT; ++sp) a += gp[sp];
            }
          }
        }
        acc[r] = a;
      }
    }
  } else if (it.type == 1) {
    // ---- a batch of alpha single links: stage their source rows (signed) and integral rows
    const int kb = it.count;  // 1 when !LDSROW
    const double pen = (g.mode == 1) ? -1.0 : -g.shift;
    const SRec rec0 = g.sa_rec[it.begin];
    const double* srow0 = C + (int64_t)rec0.src * nb;  // !LDSROW: the (unsigned) source row in place
    const double sg0 = srec_sign(rec0.meta);
    // All KM records first, then the KM rows' loads together, then the LDS writes: with one block per link
    // (record -> row -> LDS) the links of a batch were staged one memory round trip after another.  Slots
    // past the batch's count re-read link 0 and are written as zero rows (the gather loop runs over the full
    // batch capacity); only the kslots slots that exist in the LDS plan are written.
    constexpr int KM = 4;
    const int kslots = LDSROW ? g.K : 1;
    SRec recs[KM];
#pragma unroll
    for (int j = 0; j < KM; ++j) recs[j] = g.sa_rec[it.begin + (j < kb ? j : 0)];
    SCLK(k1);
    SCLK_ADD(11, k0, k1);  // type 1: item record -> alpha records (+ virtual-row header)
    if (LDSROW) vrow_records(g, pre);  // (in flight with the rows and the integral rows)
    if (tid == 0) {
#pragma unroll
      for (int j = 0; j < KM; ++j)
        if (j < kslots) penw[j] = (j < kb) ? ((int)srec_widx(recs[j].meta) ^ 1) : -1;  // same pair, opposite direction
    }
    if (LDSROW) {
      for (int64_t i = tid; i < nb; i += T) {
        double t[KM];
#pragma unroll
        for (int j = 0; j < KM; ++j) t[j] = C[(int64_t)recs[j].src * nb + i];
#pragma unroll
        for (int j = 0; j < KM; ++j)
          if (j < kslots) Crow[(int64_t)j * g.nb_pad + i] = (j < kb) ? srec_sign(recs[j].meta) * t[j] : 0.0;
      }
    }
    for (int i = tid; i < nnorb; i += T) {
      double w[KM];
#pragma unroll
      for (int j = 0; j < KM; ++j) w[j] = g.eri_pp[(int64_t)(srec_widx(recs[j].meta) >> 1) * nnorb + i];
#pragma unroll
      for (int j = 0; j < KM; ++j)
        if (j < kslots) W2[(int64_t)j * w2s + i] = (j < kb && g.mode == 0) ? w[j] : 0.0;
    }
    __syncthreads();
    SCLK(k2);
    SCLK_ADD(12, k1, k2);  // type 1: rows + integral rows + beta records staged
    const int s1 = (vs0 + g.nvs_max < vs1) ? vs0 + g.nvs_max : vs1;
    if (LDSROW) {
      for (int v = vs0 + tid; v < s1; v += T)
        part_s[v - vs0] = vrow_singles_batch_k<SPIN>(g, g.K, v, Crow, W2, w2s, penw, pen, pre, v == vs0 + tid);
    } else {
      for (int v = vs0 + tid; v < s1; v += T)
        part_s[v - vs0] = sg0 * vrow_singles_batch<SPIN, 1>(g, v, srow0, W2, w2s, penw, pen, pre, false);
    }
    __syncthreads();
#ifdef SQD_PHASE_CLOCK
    SCLK(k3);
    SCLK_ADD(13, k2, k3);  // type 1: virtual rows (LDS gathers)
    if (threadIdx.x == 0) sqd_clk_sigma[sclk_row() * SCLK_COLS + 8] += 1ull;
    sclk_mid = k3;
#endif
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int64_t B = B0 + tid + (int64_t)r * T;
      if (B < Bend) {
        double a = 0.0;
        if (g.mode == 0) {
          double jb[KM];
#pragma unroll
          for (int j = 0; j < KM; ++j) jb[j] = g.jbT[(int64_t)(srec_widx(recs[j].meta) >> 1) * nb + B];
#pragma unroll
          for (int j = 0; j < KM; ++j)
            if (j < kslots) a += (j < kb) ? jb[j] * (LDSROW ? Crow[(int64_t)j * g.nb_pad + B] : sg0 * srow0[B]) : 0.0;
        }
        a += own_rows_sum(g.vs_own, B, part_s, vs0, s1);
        acc[r] = a;
      }
    }
  } else {
    // ---- a chunk of same-spin alpha links (singles' one-body part and doubles): row AXPYs
    if (g.mode == 0) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int64_t B = B0 + tid + (int64_t)r * T;
        if (B < Bend) {
          acc[r] = axpy_chunk(C, g.ha_src, g.ha_val, it.begin, it.count, nb, B);
        }
      }
    }
  }
  double* __restrict__ out = (it.slot < 0) ? (sigma_out + (A - g.row0) * nb) : (g.partial + (int64_t)it.slot * nb);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int64_t B = B0 + tid + (int64_t)r * T;
    if (B < Bend) __builtin_nontemporal_store(acc[r], &out[B]);
  }
#ifdef SQD_PHASE_CLOCK
  {
    SCLK(k9);
    if (it.type == 0) SCLK_ADD(3, sclk_mid, k9);        // type 0: diagonal, sums of the virtual rows, (dense partials,) store
    else if (it.type == 1) SCLK_ADD(14, sclk_mid, k9);  // type 1: beta-occupation term, sums, store
    if (it.type == 0) SCLK_ADD(4, k0, k9);
    else if (it.type == 1) SCLK_ADD(15, k0, k9);
  }
#endif
  // ---- extra passes (staged rows of ~10^4 strings: one partial sum per virtual row does not fit beside
  // the row).  The staged rows are still in LDS; every further pass evaluates the next nvs_max / nvd_max
  // virtual rows and each thread adds its strings' share to the element it has just written (same thread,
  // same address: program order).  Full rows precede tails in the row order and a string's full rows are
  // contiguous, so the order of additions is fixed by the layout alone.
  if (PASS && LDSROW && it.type != 2 && ((g.type_mask >> it.type) & 1)) {
    const bool own = (it.type == 0);
    const int npass = own ? (g.mode == 0 ? npasses(vs1 - vs0, g.nvs_max, vd1 - vd0, g.nvd_max) : 1)
                          : npasses(vs1 - vs0, g.nvs_max, 0, 1);
    const double pen = (g.mode == 1) ? -1.0 : -g.shift;
    for (int ps = 1; ps < npass; ++ps) {
      const int s0 = vs0 + ps * g.nvs_max, s1 = (s0 + g.nvs_max < vs1) ? s0 + g.nvs_max : vs1;
      const int d0 = vd0 + ps * g.nvd_max, d1 = (d0 + g.nvd_max < vd1) ? d0 + g.nvd_max : vd1;
      __syncthreads();  // the previous pass's sums have been consumed
      if (own) {
        for (int v = s0 + tid; v < s1; v += T) part_s[v - s0] = vrow_singles_own(g, v, Crow, W2, g.gdense == nullptr);
        for (int v = d0 + tid; v < d1; v += T) part_d[v - d0] = vrow_doubles_own(g, v, Crow);
      } else {
        for (int v = s0 + tid; v < s1; v += T)
          part_s[v - s0] = vrow_singles_batch_k<SPIN>(g, g.K, v, Crow, W2, w2s, penw, pen, pre, false);
      }
      __syncthreads();
#pragma nounroll
      for (int r = 0; r < R; ++r) {
        const int64_t B = B0 + tid + (int64_t)r * T;
        if (B < Bend) {
          double t = own_rows_sum(g.vs_own, B, part_s, s0, s1);
          if (own) t += own_rows_sum(g.vd_own, B, part_d, d0, d1);
          out[B] += t;
        }
      }
    }
  }
}

template <int R, bool SPIN, bool LDSROW, bool PASS>
__global__ __launch_bounds__(1024) void k_sigma(const SigmaArgs g) {
  HIP_DYNAMIC_SHARED(double, smem)
  sigma_body<R, SPIN, LDSROW, PASS>(g, smem, blockIdx.x, blockIdx.y);
}
// batched (sqd_solve_batch): blockIdx.z = subspace of this launch class, arguments in device memory
template <int R, bool SPIN, bool LDSROW, bool PASS>
__global__ __launch_bounds__(1024) void k_sigma_b(const SigmaArgs* __restrict__ gs) {
  HIP_DYNAMIC_SHARED(double, smem)
  const SigmaArgs g = gs[blockIdx.z];  // (a by-value copy: the record in SGPRs, as a kernel argument would be)
  if (blockIdx.x >= g.gx || blockIdx.y >= g.gy) return;
  sigma_body<R, SPIN, LDSROW, PASS>(g, smem, blockIdx.x, blockIdx.y);
}

// ---- "direct" sigma for ultra-sparse coupling (about one in-set link per string or fewer: uniform-random string
// sets of up to ~1000 strings per spin, BASELINE's headline configuration).  There is nothing to stage or balance
// then: one thread per output element gathers its handful of contributions straight from the CSR link lists,
//   sigma[A,B] = hdiag C[A,B] + sum_{beta links of B} (...) C[A,B'] + sum_{alpha links of A} (...) C[A',B]
//              + sum_{alpha singles of A} sum_{beta singles of B} s t (pq|rs) C[A',B'],
// most loops being empty.  The work-item kernel above pays ~6 dependent memory round trips per row for its LDS
// staging and virtual-row bookkeeping whether or not a row has links (9.7 us for 319 workgroups at 317 x 317);
// this one is a single pass.  Same fixed summation order on every run (bitwise reproducible).
template <bool SPIN>
__device__ inline void sigma_direct_body(const DirectArgs& g, unsigned bx, unsigned nbx) {
  if (g.stop && *g.stop) return;
  const int64_t vsel = g.vec_index ? (int64_t)(*g.vec_index - 1) : 0;
  const double* __restrict__ C = g.c + vsel * g.c_stride;
  double* __restrict__ out = g.sigma + vsel * g.s_stride;
  const int64_t n = (g.row1 - g.row0) * g.nb;
  const double pen = (g.mode == 1) ? -1.0 : -g.shift;
  for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < n; i += (int64_t)nbx * blockDim.x)
    out[i] = direct_element<SPIN>(g, C, i, pen);
}
template <bool SPIN>
__global__ void k_sigma_direct(const DirectArgs g) {
  sigma_direct_body<SPIN>(g, blockIdx.x, gridDim.x);
}
template <bool SPIN>
__global__ void k_sigma_direct_b(const DirectArgs* __restrict__ gs) {
  const DirectArgs g = gs[blockIdx.z];  // (a by-value copy: the record in SGPRs, as a kernel argument would be)
  if (blockIdx.x >= g.gx) return;
  sigma_direct_body<SPIN>(g, blockIdx.x, g.gx);
}

// ---- Long rows, short lists (uniform-random string sets from ~10^3 strings per spin up to rows of 19 968 strings).
// The work-item kernel spends such a sigma on BYTES THROUGH THE VECTOR L1 (profiles/r01/sigma_tuning_notes.txt 10:
// 2.8 MB per 80 KB row of output at 10^4 x 10^4 -- 1.3 MB of it the beta link records, which are the same for every
// row -- at ~50 GB/s per CU, plus a pass that adds partial rows).  Here a workgroup owns R whole rows of C:
//  * the R rows are staged in LDS (R = 2 at nb = 10^4: 156 KB), so every beta record read serves R rows and its
//    operands are LDS gathers;
//  * a lane owns a target column B and walks ALL of B's links itself: the sums stay in registers -- no virtual rows,
//    no partial sums in LDS, no partial rows in memory, no reduce launch.  That is only sound because the lists are
//    short and even (the selection rule in build_subspace); Hartree-Fock-centred sets keep the work-item kernel;
//  * the beta doubles are read in per-slice jagged-diagonal order (k_tables_jds): at step k the lanes of a wavefront
//    whose list is longer than k read consecutive records -- coalesced and without padding; positions come from a
//    ballot, no index array;
//  * the alpha lists are wave-uniform (scalar loads) and their source rows are read coalesced, eight in flight.
// The order of accumulation per element is k_sigma_direct's, so the two kernels agree to the bit.
template <int R, bool SPIN>
__device__ inline void sigma_rows_body(const DirectArgs& g, double* srow, unsigned bx) {
  // J slices (of 64 columns) in flight per wavefront, K links per round: a workgroup that fills the CU's LDS runs
  // alone on it, so nothing but its own instruction stream hides the latency of a chain pointers -> records ->
  // operands.  One slice at a time that chain was ~7 round trips per slice and the skeleton alone (no links at all)
  // took 1.8 ms at 10^4 x 10^4; four independent chains interleaved cut it by the same factor.
  constexpr int J = (R >= 8) ? 1 : (R >= 3) ? 2 : 4, K = 4;  // (registers: J R sums + J K operands under 128 VGPRs)
  if (g.stop && *g.stop) return;
  const int64_t vsel = g.vec_index ? (int64_t)(*g.vec_index - 1) : 0;
  const double* __restrict__ C = g.c + vsel * g.c_stride;
  double* __restrict__ out = g.sigma + vsel * g.s_stride;
  const int64_t nb = g.nb, pitch = g.nb_pad;
  const int64_t A0 = g.row0 + (int64_t)bx * R;
  const int nr = (int)((g.row1 - A0) < R ? (g.row1 - A0) : R);  // rows of this workgroup (the last one may be short)
  const double pen = (g.mode == 1) ? -1.0 : -g.shift;
  {
    // staging: eight loads per thread in flight, then the eight LDS stores (element by element the loop was a chain of
    // ~10 dependent round trips per row for a workgroup that has the CU to itself)
    const int64_t T8 = (int64_t)blockDim.x * 8;
    for (int r = 0; r < nr; ++r) {
      const double* __restrict__ crow = C + (A0 + r) * nb;
      for (int64_t b0 = threadIdx.x; b0 < nb; b0 += T8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t b = b0 + (int64_t)u * blockDim.x;
          v[u] = crow[b < nb ? b : b0];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int64_t b = b0 + (int64_t)u * blockDim.x;
          if (b < nb) srow[r * pitch + b] = v[u];
        }
      }
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int64_t T = blockDim.x;
  for (int64_t Bw = (int64_t)(threadIdx.x >> 6) * 64; Bw < nb; Bw += T * J) {
    // slice j of this pass: columns Bw + j T .. + 63.  Dead lanes (and dead slices) shadow the last column, have
    // empty lists and store nothing.
    int64_t Bc[J];
    bool live[J];
    int len[J];
    int64_t base[J], slice0[J], sb0[J], sb1[J];
    double acc[J][R];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int64_t B = Bw + j * T + lane;
      live[j] = B < nb;
      Bc[j] = live[j] ? B : nb - 1;
      const int64_t first = (Bw + j * T < nb) ? Bw + j * T : nb - 1;
      const int64_t d0 = g.db_ptr[Bc[j]], d1 = g.db_ptr[Bc[j] + 1];
      len[j] = live[j] ? (int)(d1 - d0) : 0;
      slice0[j] = g.db_ptr[first];
      base[j] = slice0[j];
      sb0[j] = g.sb_ptr[Bc[j]];
      sb1[j] = live[j] ? g.sb_ptr[Bc[j] + 1] : sb0[j];
      const uint64_t sB = g.strs_b[Bc[j]];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int rr = r < nr ? r : 0;
        const int64_t A = A0 + rr;
        const double cv = srow[rr * pitch + Bc[j]];
        if (g.mode == 0) {
          double d = g.hdiag[(A - g.row0) * nb + Bc[j]];
          if (SPIN) d += g.shift * (g.szterm + (double)__popcll(sB & ~g.strs_a[A]) - g.ss);
          acc[j][r] = d * cv;
        } else {
          acc[j][r] = (g.szterm + (double)__popcll(sB & ~g.strs_a[A])) * cv;
        }
      }
    }
    if (g.mode == 0) {
      // beta singles (CSR: rare in this regime): same-spin value + alpha occupation term
#pragma unroll
      for (int j = 0; j < J; ++j)
        for (int64_t l = sb0[j]; l < sb1[j]; ++l) {
          const SRec rec = g.sb_rec[l];
          const double v = g.sb_val[l], sg = srec_sign(rec.meta);
          const int64_t jw = srec_widx(rec.meta) >> 1;
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (r < nr) acc[j][r] += (v + sg * g.ja_row[(A0 + r) * g.nnorb + jw]) * srow[r * pitch + rec.src];
        }
      // beta doubles, jagged-diagonal order of each slice: the K positions of a round come from K ballots, then the
      // 2 J K loads are in flight together.  Lanes past the end of their list re-read the slice's first record with
      // weight zero.
      for (int k0 = 0;; k0 += K) {
        bool more = false;
#pragma unroll
        for (int j = 0; j < J; ++j) more = more || (k0 < len[j]);
        if (!__ballot(more)) break;
        uint32_t src[J][K];
        double v[J][K];
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
          for (int u = 0; u < K; ++u) {
            const bool on = k0 + u < len[j];
            const unsigned long long m = __ballot(on);
            const int64_t pos = on ? base[j] + __popcll(m & lt) : slice0[j];
            base[j] += __popcll(m);
            src[j][u] = g.jd_src[pos];
            const double val = g.jd_val[pos];
            v[j][u] = on ? val : 0.0;
          }
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
          for (int u = 0; u < K; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r)
              if (r < nr) acc[j][r] += v[j][u] * srow[r * pitch + src[j][u]];
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (r >= nr) continue;  // (uniform over the workgroup)
      const int64_t A = A0 + r;
      const int64_t sa0 = g.sa_ptr[A], sa1 = g.sa_ptr[A + 1];
      if (g.mode == 0) {
        // alpha same-spin links (wave-uniform lists, coalesced source rows): singles' one-body part, then doubles,
        // K source rows x J slices in flight per round; the padding of the last round re-reads the list's last link
        // (the same cache lines again) with weight zero
        for (int64_t l = sa0; l < sa1; ++l) {
          const double w = g.sa_val[l];
          const double* __restrict__ srcrow = C + (int64_t)g.sa_rec[l].src * nb;
#pragma unroll
          for (int j = 0; j < J; ++j) acc[j][r] += w * srcrow[Bc[j]];
        }
        const int64_t da0 = g.da_ptr[A], da1 = g.da_ptr[A + 1];
        for (int64_t l = da0; l < da1; l += K) {
          double x[J][K], w[K];
#pragma unroll
          for (int u = 0; u < K; ++u) {
            const bool on = l + u < da1;
            const int64_t lu = on ? l + u : da1 - 1;
            w[u] = on ? g.da_val[lu] : 0.0;
            const double* __restrict__ srcrow = C + (int64_t)g.da_src[lu] * nb;
#pragma unroll
            for (int j = 0; j < J; ++j) x[j][u] = srcrow[Bc[j]];
          }
#pragma unroll
          for (int j = 0; j < J; ++j)
#pragma unroll
            for (int u = 0; u < K; ++u) acc[j][r] += w[u] * x[j][u];
        }
        // alpha singles x beta occupation
        for (int64_t ls = sa0; ls < sa1; ++ls) {
          const SRec rec = g.sa_rec[ls];
          const double sg = srec_sign(rec.meta);
          const double* __restrict__ jrow = g.jbT + (int64_t)(srec_widx(rec.meta) >> 1) * nb;
          const double* __restrict__ srcrow = C + (int64_t)rec.src * nb;
#pragma unroll
          for (int j = 0; j < J; ++j) acc[j][r] += sg * jrow[Bc[j]] * srcrow[Bc[j]];
        }
      }
      // single x single (and the S^2 exchange term: the beta link that undoes the alpha link's orbital move)
      for (int64_t la = sa0; la < sa1; ++la) {
        const SRec ra = g.sa_rec[la];
        const double* srcrow = C + (int64_t)ra.src * nb;
        const double* w = g.eri_pp + (int64_t)(srec_widx(ra.meta) >> 1) * g.nnorb;
        const int partner = (int)srec_widx(ra.meta) ^ 1;
#pragma unroll
        for (int j = 0; j < J; ++j) {
          double t = 0.0;
          for (int64_t lb = sb0[j]; lb < sb1[j]; ++lb) {
            const SRec rb = g.sb_rec[lb];
            double wv = (g.mode == 0) ? w[srec_widx(rb.meta) >> 1] : 0.0;
            if (SPIN) wv += ((int)srec_widx(rb.meta) == partner) ? pen : 0.0;
            t += srec_sign(rb.meta) * wv * srcrow[rb.src];
          }
          acc[j][r] += srec_sign(ra.meta) * t;
        }
      }
#pragma unroll
      for (int j = 0; j < J; ++j)
        if (live[j]) out[(A - g.row0) * nb + Bw + j * T + lane] = acc[j][r];
    }
  }
}

template <int R, bool SPIN>
__global__ void __launch_bounds__(1024) k_sigma_rows(const DirectArgs g) {
  HIP_DYNAMIC_SHARED(double, srow)
  sigma_rows_body<R, SPIN>(g, srow, blockIdx.x);
}
template <int R, bool SPIN>
__global__ void __launch_bounds__(1024) k_sigma_rows_b(const DirectArgs* __restrict__ gs) {
  HIP_DYNAMIC_SHARED(double, srow)
  const DirectArgs g = gs[blockIdx.z];  // (a by-value copy: the record in SGPRs, as a kernel argument would be)
  if (blockIdx.x >= g.gx) return;
  sigma_rows_body<R, SPIN>(g, srow, blockIdx.x);
}

// sigma[A,:] = sum over the partial rows of A (fixed order) for rows that were split into several items.
// Workgroup = 64 columns x SL slot lanes: lane sl adds slots sl, sl+SL, ... (independent loads), the SL
// partial sums meet in LDS and are added in slot-lane order => bitwise reproducible.
// ---- same-spin part on the f64 matrix cores (dense same-spin mode):  G = H_a C + C H_b  with H_a (na x na), H_b
// (nb x nb) the dense, zero-padded, SYMMETRIC same-spin blocks (k_tables_dense) and C the na x nb vector.  MFMA is used
// here because the blocks of the sets the SQD loop produces are 20-22 % dense: enumerating the links costs two LDS
// gathers per multiply-add, the dense product none, and 5 x the flops on v_mfma_f64_16x16x4_f64 are cheaper.
// One workgroup (4 wavefronts) = one 64 x 64 tile of G; wavefront w owns the 32 x 32 quadrant (w / 2, w % 2) as 2 x 2
// MFMA tiles (16 f64 accumulators per lane).  Both products run through the same loop over chunks of DK = 16 k: the 64 x 16
// operand tiles are staged in LDS k-major -- H_a[i][k] is read as H_a[k][i] (symmetric: coalesced), the rows of C for
// the second product are transposed on the way in (pitch 65) -- so every fragment read is 16 consecutive doubles.
// Register-staged double buffering: the global loads of chunk c + 1 are in flight while chunk c is multiplied.
// Fixed order of accumulation: the same bits on every run, and in a batched launch.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
struct DenseArgs {
  GPtr<const double> c;   // the vector(s): na x nb, row-major
  GPtr<double> g;         // the product, na x nb
  GPtr<const double> ha, hb;
  int64_t na, nb;
  int pa, pb;             // leading dimensions (multiples of 64)
  GPtr<const int> stop, vec_index;
  int64_t c_stride;
  unsigned tj;            // tiles along B
  unsigned gx;            // tiles of this subspace (x DENSE_SPLIT workgroups: blockIdx.y = k range)
};
constexpr int DT = 64, DK = 16, DU = DT * DK / 256;  // DU: elements per thread and operand tile
// LDS layouts, chosen so that every fragment read (ds_read_b64: lane groups {0-31}, {32-63}, bank = word mod 64) and every
// staging write is conflict-free: k-major tiles [k][64] at a pitch of 80 doubles (the two k rows of a lane group start
// 32 banks apart), the rows of C of product 2 as they are, [i][16 k] at a pitch of 18 doubles (16 rows x 36 words
// are 16 distinct multiples of 4 mod 64; the group's second k sits 2 banks further)
constexpr int DPITCH = 80, DPITCH2 = DK + 2, DTILE = DK * DPITCH;  // DTILE doubles per staged tile (>= 64 * DPITCH2)
__device__ inline void same_spin_mfma_body(const DenseArgs& g, unsigned bx, unsigned by) {
  // (declared HERE and addressed by integer offsets: through a `double*` parameter or an array of buffer pointers the
  // compiler loses the address space and emits flat_load / flat_store for the tiles, and a flat access waits on the
  // vector-memory counter too -- the prefetch of the next chunk was waited for before the MFMAs it should hide behind)
  __shared__ double smem[4 * DTILE];
  if (g.stop && *g.stop) return;
  const int64_t vsel = g.vec_index ? (int64_t)(*g.vec_index - 1) : 0;
  const double* __restrict__ C = g.c + vsel * g.c_stride;
  const double* __restrict__ Ha = g.ha;
  const double* __restrict__ Hb = g.hb;
  // (orders up to 4096: every element offset fits 32 bits)
  const int na = (int)g.na, nb = (int)g.nb, pa = g.pa, pb = g.pb;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int wi = wv >> 1, wj = wv & 1;
  const int i0 = (int)(bx / g.tj) * DT, j0 = (int)(bx % g.tj) * DT;
  // buffer `buf`: operand tile A at 2 buf DTILE, operand tile B at (2 buf + 1) DTILE
  // chunk list: product 1 over k in [0, na16), then product 2 over k in [0, nb16); this workgroup's share of it
  // (by = which of the DENSE_SPLIT partial products)
  const int n1 = (na + DK - 1) / DK, n2 = (nb + DK - 1) / DK, nch = n1 + n2;
  const int ch0 = (int)((int64_t)nch * by / DENSE_SPLIT), ch1 = (int)((int64_t)nch * (by + 1) / DENSE_SPLIT);
  // this thread's DU + DU elements of a chunk's operand tiles: tile element e = t + 256 u is (kk, cc) = (e / 64, e % 64)
  // for the k-major reads, (ii, k2) = (e / DK, e % DK) for the rows of C of product 2
  const int kk0 = t >> 6, cc = t & 63, ii0 = t / DK, k2 = t % DK;
  const bool col_ok = j0 + cc < nb;
  double ra[DU], rb[DU];
  auto fetch = [&](int ch) {
    if (ch < n1) {
      const int k0 = ch * DK;
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const int kr = k0 + kk0 + (256 / 64) * u;
        ra[u] = Ha[kr * pa + i0 + cc];                                   // H_a[i0+cc][kr] (symmetric: read k-major)
        rb[u] = (kr < na && col_ok) ? C[kr * nb + j0 + cc] : 0.0;        // C[kr][j0+cc]
      }
    } else {
      const int k0 = (ch - n1) * DK;
#pragma unroll
      for (int u = 0; u < DU; ++u) {
        const int ir = i0 + ii0 + (256 / DK) * u, kc = k0 + k2;
        ra[u] = (ir < na && kc < nb) ? C[ir * nb + kc] : 0.0;            // C[i0+ii][k0+k2]
        rb[u] = Hb[(k0 + kk0 + (256 / 64) * u) * pb + j0 + cc];          // H_b[kr][j0+cc]
      }
    }
  };
  auto park = [&](int ch, int buf) {  // registers -> LDS
    const int oa = 2 * buf * DTILE, ob = oa + DTILE;
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      if (ch < n1) smem[oa + (kk0 + (256 / 64) * u) * DPITCH + cc] = ra[u];      // k-major
      else smem[oa + (ii0 + (256 / DK) * u) * DPITCH2 + k2] = ra[u];             // rows of C as they are
      smem[ob + (kk0 + (256 / 64) * u) * DPITCH + cc] = rb[u];
    }
  };
  mfma_d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  if (ch0 < ch1) {
    fetch(ch0);
    park(ch0, ch0 & 1);
  }
  __syncthreads();
  const int fa1 = lk * DPITCH + wi * 32 + li, fa2 = (wi * 32 + li) * DPITCH2 + lk, fb = lk * DPITCH + wj * 32 + li;
  for (int ch = ch0; ch < ch1; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < ch1) fetch(ch + 1);
    const bool p1 = ch < n1;
    const int oa = 2 * buf * DTILE + (p1 ? fa1 : fa2), ob = (2 * buf + 1) * DTILE + fb;
    const int ak = p1 ? 4 * DPITCH : 4, a16 = p1 ? 16 : 16 * DPITCH2;  // fragment strides: next k4 step, next 16 rows
#pragma unroll
    for (int k4 = 0; k4 < DK / 4; ++k4) {
      const double a0 = smem[oa + k4 * ak], a1 = smem[oa + k4 * ak + a16];
      const double b0 = smem[ob + k4 * 4 * DPITCH], b1 = smem[ob + k4 * 4 * DPITCH + 16];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (ch + 1 < ch1) park(ch + 1, buf ^ 1);
    __syncthreads();
  }
  double* __restrict__ G = g.g + (int64_t)by * g.na * g.nb;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = i0 + wi * 32 + a * 16 + lk + 4 * r, col = j0 + wj * 32 + b * 16 + li;
        if (row < na && col < nb) G[row * nb + col] = acc[a][b][r];
      }
}
// (40 KB of LDS and at most 128 VGPRs: three to four workgroups per CU, so that one workgroup's loads hide behind the
// others' MFMAs)
__global__ void __launch_bounds__(256, 4) k_same_spin_mfma(const DenseArgs g) {
  same_spin_mfma_body(g, blockIdx.x, blockIdx.y);
}
__global__ void __launch_bounds__(256, 4) k_same_spin_mfma_b(const DenseArgs* __restrict__ gs) {
  const DenseArgs g = gs[blockIdx.z];  // (a by-value copy: the record in SGPRs, as a kernel argument would be)
  if (blockIdx.x >= g.gx) return;
  same_spin_mfma_body(g, blockIdx.x, blockIdx.y);
}

struct ReduceArgs {
  GPtr<const MultiRow> rows;
  GPtr<const double> partial;
  int64_t nb;
  GPtr<double> sigma;
  GPtr<const int> stop;
  GPtr<const int> vec_index;
  int64_t s_stride, row0;
  unsigned gx, gy;
};
__device__ inline void sigma_reduce_body(const ReduceArgs& g, unsigned bx, unsigned by) {
  __shared__ double red[1024];
  const MultiRow* __restrict__ rows = g.rows;
  const double* __restrict__ partial = g.partial;
  const int64_t nb = g.nb, row0 = g.row0;
  double* __restrict__ sigma = g.sigma;
  if (g.stop && *g.stop) return;
  if (g.vec_index) sigma += (int64_t)(*g.vec_index - 1) * g.s_stride;
  const MultiRow mr = rows[bx];
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6, SL = blockDim.x >> 6;
  const int64_t B = (int64_t)by * 64 + col;
  double s = 0.0;
  if (B < nb)
    for (int j = sl; j < mr.nslots; j += SL) s += partial[(int64_t)(mr.slot0 + j) * nb + B];
  red[threadIdx.x] = s;
  __syncthreads();
  if (sl == 0 && B < nb) {
    for (int r = 1; r < SL; ++r) s += red[r * 64 + col];
    sigma[((int64_t)mr.A - row0) * nb + B] = s;
  }
}
__global__ void k_sigma_reduce(const ReduceArgs g) { sigma_reduce_body(g, blockIdx.x, blockIdx.y); }
__global__ void k_sigma_reduce_b(const ReduceArgs* __restrict__ gs) {
  const ReduceArgs g = gs[blockIdx.z];  // (a by-value copy: the record in SGPRs, as a kernel argument would be)
  if (blockIdx.x >= g.gx || blockIdx.y >= g.gy) return;
  sigma_reduce_body(g, blockIdx.x, blockIdx.y);
}

// y = a*x + b*y   (x / y optionally selected on the device like the sigma vectors: stride 0 = not indexed)
__global__ void k_axpby(int64_t n, double a, const double* __restrict__ x, double b, double* __restrict__ y,
                        const int* stop, const int* vec_index, int64_t x_stride, int64_t y_stride) {
  if (stop && *stop) return;
  if (vec_index) {
    x += (int64_t)(*vec_index - 1) * x_stride;
    y += (int64_t)(*vec_index - 1) * y_stride;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = a * x[i] + b * y[i];
}

template <int R, bool SPIN, bool LDSROW, bool PASS>
static int launch_sigma_rs(sqd_ctx* c, const SigmaArgs& g) {
  if (c->sig_shmem > 64 * 1024) {
    // once per (instantiation, device) and size step, not per launch
    static std::atomic<size_t> granted[64];
    const int dev = c->device & 63;
    if (c->sig_shmem > granted[dev].load(std::memory_order_relaxed)) {
      SQD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sigma<R, SPIN, LDSROW, PASS>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->sig_shmem));
      granted[dev].store(c->sig_shmem, std::memory_order_relaxed);
    }
  }
  hipLaunchKernelGGL((k_sigma<R, SPIN, LDSROW, PASS>), dim3((unsigned)c->n_items, (unsigned)c->sig_nchunks), dim3(c->sig_T),
                     c->sig_shmem, c->stream, g);
  SQD_HIP_CHECK(hipGetLastError());
  if (c->ev_after_sigma_kernel) {  // profiling: duration of k_sigma alone (the reduce follows)
    SQD_HIP_CHECK(hipEventRecord(c->ev_after_sigma_kernel, c->stream));
    c->ev_after_sigma_kernel = nullptr;
  }
  return SQD_OK;
}
template <int R>
static int launch_sigma_r(sqd_ctx* c, const SigmaArgs& g) {
  const bool spin = (g.mode == 1 || g.spin);
  // multi-pass walk of the beta lists -- also for rows of more than eight columns per thread (R = 16) whose lists fit one
  // pass: k_sigma<16, ., true, false> does not return on the MI355X (profiles/r05/long_rows_hang_probe.txt) while the
  // multi-pass instantiation, here with zero extra passes, is the one every set of ~10^4 strings has run since round 2
  static const bool r16_single = [] {  // probe / test hook: rows of more than eight columns per thread in the single-pass form
    const char* env = std::getenv("SQD_SIGMA_R16_SINGLE");
    return env && std::atoi(env) != 0;
  }();
  if ((R > 8 && !r16_single) || c->sig_ps < c->hv_s.nv_max || c->sig_pd < c->hv_d.nv_max)
    return spin ? launch_sigma_rs<R, true, true, true>(c, g) : launch_sigma_rs<R, false, true, true>(c, g);
  return spin ? launch_sigma_rs<R, true, true, false>(c, g) : launch_sigma_rs<R, false, true, false>(c, g);
}
// rows in global memory: R is 1 (test hook) or 4 (4096-column chunks of 1024 threads)
template <int R>
static int launch_sigma_g(sqd_ctx* c, const SigmaArgs& g) {
  return (g.mode == 1 || g.spin) ? launch_sigma_rs<R, true, false, false>(c, g)
                                 : launch_sigma_rs<R, false, false, false>(c, g);
}

// workgroups of the element-gather / whole-rows kernel on this subspace
static unsigned direct_blocks(const sqd_ctx* c) {
  if (c->sig_rows > 0) return (unsigned)((c->row1 - c->row0 + c->sig_rows - 1) / c->sig_rows);
  const int64_t n = (c->row1 - c->row0) * c->nb;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  return (unsigned)blocks;
}
void fill_direct_args(sqd_ctx* c, const double* d_c, double* d_sigma, int mode, bool spin, double ss, double shift,
                      int64_t in_stride, int64_t out_stride, DirectArgs* gp) {
  const SpinTables& a = c->sp[0];
  const SpinTables& b = c->sp[1];
  DirectArgs& g = *gp;
  g.c = d_c;
  g.sigma = d_sigma;
  g.hdiag = c->hdiag.as<double>();
  g.row0 = c->row0;
  g.row1 = c->row1;
  g.nb = c->nb;
  g.nnorb = c->nnorb;
  g.mode = mode;
  g.spin = spin ? 1 : 0;
  g.ss = ss;
  g.shift = shift;
  const double sz = 0.5 * (c->nelec[0] - c->nelec[1]);
  g.szterm = sz * (sz + 1.0);
  g.strs_a = a.strs.as<uint64_t>();
  g.strs_b = b.strs.as<uint64_t>();
  g.sa_ptr = a.s_ptr.as<int64_t>();
  g.da_ptr = a.d_ptr.as<int64_t>();
  g.sb_ptr = b.s_ptr.as<int64_t>();
  g.db_ptr = b.d_ptr.as<int64_t>();
  g.sa_rec = a.s_rec.as<SRec>();
  g.sb_rec = b.s_rec.as<SRec>();
  g.sa_val = a.s_val.as<double>();
  g.sb_val = b.s_val.as<double>();
  g.da_src = a.d_src.as<uint32_t>();
  g.da_val = a.d_val.as<double>();
  g.db_src = b.d_src.as<uint32_t>();
  g.db_val = b.d_val.as<double>();
  g.ja_row = a.jrow.as<double>();
  g.jbT = b.jT.as<double>();
  g.eri_pp = c->eri_pp.as<double>();
  g.stop = c->sigma_stop;
  const bool indexed = c->sigma_index && (in_stride || out_stride);
  g.vec_index = indexed ? c->sigma_index : nullptr;
  g.c_stride = in_stride;
  g.s_stride = out_stride;
  g.jd_src = b.jd_src.as<uint32_t>();
  g.jd_val = b.jd_val.as<double>();
  g.nb_pad = (c->nb + 1) & ~int64_t(1);
  g.gx = direct_blocks(c);
}

static int launch_sigma_direct(sqd_ctx* c, const double* d_c, double* d_sigma, int mode, bool spin, double ss,
                               double shift, int64_t in_stride, int64_t out_stride) {
  DirectArgs g;
  fill_direct_args(c, d_c, d_sigma, mode, spin, ss, shift, in_stride, out_stride, &g);
  if (c->sig_rows > 0) {
    const int R = c->sig_rows;
    const size_t shmem = (size_t)R * g.nb_pad * 8;
    const unsigned blocks_r = g.gx;
    static const int T = [] {  // tuning hook
      const char* env = std::getenv("SQD_ROWS_T");
      const int v = env ? std::atoi(env) : 1024;
      return (v >= 64 && v <= 1024 && v % 64 == 0) ? v : 1024;
    }();
    const bool sp = (mode == 1 || spin);
#define SQD_ROWS_LAUNCH(RR, SS)                                                                                  \
  do {                                                                                                           \
    if (shmem > 64 * 1024) {                                                                                     \
      static std::atomic<size_t> granted[64];                                                                    \
      const int dev = c->device & 63;                                                                            \
      if (shmem > granted[dev].load(std::memory_order_relaxed)) {                                                \
        SQD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sigma_rows<RR, SS>),                  \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));              \
        granted[dev].store(shmem, std::memory_order_relaxed);                                                    \
      }                                                                                                          \
    }                                                                                                            \
    hipLaunchKernelGGL((k_sigma_rows<RR, SS>), dim3(blocks_r), dim3(T), shmem, c->stream, g);                    \
  } while (0)
    switch (R) {
      case 1: if (sp) SQD_ROWS_LAUNCH(1, true); else SQD_ROWS_LAUNCH(1, false); break;
      case 2: if (sp) SQD_ROWS_LAUNCH(2, true); else SQD_ROWS_LAUNCH(2, false); break;
      case 3: if (sp) SQD_ROWS_LAUNCH(3, true); else SQD_ROWS_LAUNCH(3, false); break;
      case 4: if (sp) SQD_ROWS_LAUNCH(4, true); else SQD_ROWS_LAUNCH(4, false); break;
      case 6: if (sp) SQD_ROWS_LAUNCH(6, true); else SQD_ROWS_LAUNCH(6, false); break;
      default: if (sp) SQD_ROWS_LAUNCH(8, true); else SQD_ROWS_LAUNCH(8, false); break;
    }
#undef SQD_ROWS_LAUNCH
    SQD_HIP_CHECK(hipGetLastError());
    if (c->ev_after_sigma_kernel) {
      SQD_HIP_CHECK(hipEventRecord(c->ev_after_sigma_kernel, c->stream));
      c->ev_after_sigma_kernel = nullptr;
    }
    return SQD_OK;
  }
  if (mode == 1 || spin)
    hipLaunchKernelGGL((k_sigma_direct<true>), dim3(g.gx), dim3(256), 0, c->stream, g);
  else
    hipLaunchKernelGGL((k_sigma_direct<false>), dim3(g.gx), dim3(256), 0, c->stream, g);
  SQD_HIP_CHECK(hipGetLastError());
  if (c->ev_after_sigma_kernel) {
    SQD_HIP_CHECK(hipEventRecord(c->ev_after_sigma_kernel, c->stream));
    c->ev_after_sigma_kernel = nullptr;
  }
  return SQD_OK;
}

// arguments of the work-item kernel on this subspace
static void fill_sigma_args(sqd_ctx* c, const double* d_c, double* d_sigma, int mode, bool spin, double ss, double shift,
                            int64_t in_stride, int64_t out_stride, SigmaArgs* gp) {
  SigmaArgs& g = *gp;
  const SpinTables& a = c->sp[0];
  const SpinTables& b = c->sp[1];
  g.c = d_c;
  g.sigma = d_sigma;
  g.partial = c->sig_partial.as<double>();
  g.items = c->items.as<WorkItem>();
  g.na = c->na;
  g.nb = c->nb;
  g.row0 = c->row0;
  g.nnorb = c->nnorb;
  g.nb_pad = c->sig_nb_pad;
  g.K = c->sig_K;
  g.mode = mode;
  static const int type_mask = [] {  // profiling hook, read once per process
    const char* env = std::getenv("SQD_SIGMA_TYPES");
    return env ? std::atoi(env) : 7;
  }();
  g.type_mask = type_mask;
  g.spin = spin ? 1 : 0;
  g.ss = ss;
  g.shift = shift;
  const double sz = 0.5 * (c->nelec[0] - c->nelec[1]);
  g.szterm = sz * (sz + 1.0);
  g.strs_a = a.strs.as<uint64_t>();
  g.strs_b = b.strs.as<uint64_t>();
  g.hdiag = c->hdiag.as<double>();
  g.sa_rec = a.s_rec.as<SRec>();
  g.ha_src = a.hs_src.as<uint32_t>();
  g.ha_val = a.hs_val.as<double>();
  g.ja_row = a.jrow.as<double>();
  g.vs_cnt = b.vs_cnt.as<int32_t>();
  g.vs_own = b.vs_own.as<int32_t>();
  g.vd_cnt = b.vd_cnt.as<int32_t>();
  g.vd_own = b.vd_own.as<int32_t>();
  g.nv_s = (int)b.nv_s;
  g.nv_d = (int)b.nv_d;
  g.esb_sl = b.es_sl.as<int64_t>();
  g.esb_rec = b.es_rec.as<SRec>();
  g.esb_val = b.es_val.as<double>();
  g.edb_sl = b.ed_sl.as<int64_t>();
  g.edb_src = b.ed_src.as<uint32_t>();
  g.edb_val = b.ed_val.as<double>();
  g.jbT = b.jT.as<double>();
  g.eri_pp = c->eri_pp.as<double>();

  g.vs_chunk = b.vs_chunk.as<int32_t>();
  g.vd_chunk = b.vd_chunk.as<int32_t>();
  g.chunk_cols = c->sig_chunk;
  g.nvs_max = (int)c->sig_ps;
  g.nvd_max = (int)c->sig_pd;
  g.stop = c->sigma_stop;
  const bool indexed = c->sigma_index && (in_stride || out_stride);
  g.vec_index = indexed ? c->sigma_index : nullptr;
  g.c_stride = in_stride;
  g.s_stride = out_stride;
  g.T = c->sig_T;
  g.gx = (unsigned)c->n_items;
  g.gy = (unsigned)c->sig_nchunks;
  g.gdense = (c->sig_dense && mode == 0) ? c->gdense.as<double>() : nullptr;
  g.gdense_stride = c->na * c->nb;
  g.gsplit = c->sig_spmm ? 1 : DENSE_SPLIT;
  g.c_own = nullptr;
  g.own_mode = 0;
  g.mask_skip = 0;
  if (c->sig_part == 1) {  // in front of the all-gather: own-row items on the rank's own rows
    g.c_own = c->sig_c_own;
    g.own_mode = 1;
    g.type_mask = 1;
    g.mask_skip = 1;
  } else if (c->sig_part == 2) {  // behind it: the rest
    g.own_mode = 2;
    g.type_mask = 7;
    g.mask_skip = 1;
  }
}
// arguments of the matrix-core same-spin product for the vector the work items of the same sigma build will read
static void fill_dense_args(sqd_ctx* c, const double* d_c, int64_t in_stride, DenseArgs* dp) {
  DenseArgs& d = *dp;
  d.c = d_c;
  d.g = c->gdense.as<double>();
  d.ha = c->hdense_a.as<double>();
  d.hb = c->hdense_b.as<double>();
  d.na = c->na;
  d.nb = c->nb;
  d.pa = c->dense_pa;
  d.pb = c->dense_pb;
  d.stop = c->sigma_stop;
  d.vec_index = (c->sigma_index && in_stride) ? c->sigma_index : nullptr;
  d.c_stride = in_stride;
  d.tj = (unsigned)((c->nb + DT - 1) / DT);
  d.gx = d.tj * (unsigned)((c->na + DT - 1) / DT);
}
// does a work-item sigma build of this subspace need the k_sigma_reduce launch behind it?  (Inside a Davidson run the
// first reader of the new vector adds the partial rows instead.)
static bool sigma_needs_reduce(const sqd_ctx* c, int mode, bool indexed) {
  return c->n_multi > 0 && !(c->sigma_defer_reduce && indexed && mode == 0);
}
static void fill_reduce_args(sqd_ctx* c, double* d_sigma, const SigmaArgs& g, ReduceArgs* r) {
  r->rows = c->multi.as<MultiRow>();
  r->partial = c->sig_partial.as<double>();
  r->nb = c->nb;
  r->sigma = d_sigma;
  r->stop = g.stop;
  r->vec_index = g.vec_index;
  r->s_stride = g.s_stride;
  r->row0 = c->row0;
  r->gx = (unsigned)c->n_multi;
  r->gy = (unsigned)((c->nb + 63) / 64);
}

int launch_sigma(sqd_ctx* c, const double* d_c, double* d_sigma, int mode, bool spin, double ss, double shift,
                 int64_t in_stride, int64_t out_stride) {
  if (!c->have_subspace) {
    set_error("no subspace set");
    return SQD_ERR_STATE;
  }
  // (two-launch sigma of a row shard: only the work-item kernel has a part that needs no remote row; the others run whole
  // behind the gather)
  // (a context that holds ALL rows may also have chosen a dense / sparse-product same-spin formulation: whole as well)
  // (and subspaces whose beta lists need extra passes: one launch rounds ((d c + sums + axpy) + extra-pass sums), two
  // launches ((d c + sums) + extra-pass sums) + axpy -- the split promises the same bits, so those run whole too)
  const bool multi_pass = c->sig_ps < c->hv_s.nv_max || c->sig_pd < c->hv_d.nv_max;
  const bool can_split = !c->sig_lists && !c->sig_direct && !c->sig_dense && !c->sig_opp && !multi_pass;
  if (c->sig_part == 1 && !can_split) return SQD_OK;
  struct PartGuard {  // parts of a kernel that cannot be split: the second call runs everything
    sqd_ctx* c;
    int saved;
    ~PartGuard() { c->sig_part = saved; }
  } part_guard{c, c->sig_part};
  if (!can_split) c->sig_part = 0;
  if (c->sig_lists) return launch_sigma_lists(c, d_c, d_sigma, mode, spin, ss, shift, in_stride, out_stride);
  if (c->sig_direct) return launch_sigma_direct(c, d_c, d_sigma, mode, spin, ss, shift, in_stride, out_stride);
  if (c->sig_opp && mode == 0) {
    // connected sets of ~10^3 strings per spin and more, the operator with or without the linear spin penalty: same-spin
    // product, then the opposite-spin part and the diagonal by whole rows -- no work items, no partial rows
    SQD_TRY(spmm_launch(c, d_c, in_stride));
    return opp_launch(c, d_c, d_sigma, in_stride, out_stride, spin, ss, shift);
  }
  SigmaArgs g;
  fill_sigma_args(c, d_c, d_sigma, mode, spin, ss, shift, in_stride, out_stride, &g);
  if (g.gdense && c->sig_spmm) {
    SQD_TRY(spmm_launch(c, d_c, in_stride));
  } else if (g.gdense) {
    DenseArgs d;
    fill_dense_args(c, d_c, in_stride, &d);
    hipLaunchKernelGGL(k_same_spin_mfma, dim3(d.gx, DENSE_SPLIT), dim3(256), 0, c->stream, d);
    SQD_HIP_CHECK(hipGetLastError());
  }
  const int R = c->sig_R;
  int rc;
  if (!c->sig_lds_rows) rc = (R <= 1) ? launch_sigma_g<1>(c, g) : launch_sigma_g<4>(c, g);
  else if (R <= 1) rc = launch_sigma_r<1>(c, g);
  else if (R <= 2) rc = launch_sigma_r<2>(c, g);
  else if (R <= 4) rc = launch_sigma_r<4>(c, g);
  else if (R <= 8) rc = launch_sigma_r<8>(c, g);
  else rc = launch_sigma_r<16>(c, g);
  if (rc != SQD_OK) return rc;
  if (c->sig_part != 1 && sigma_needs_reduce(c, mode, g.vec_index != nullptr)) {
    ReduceArgs r;
    fill_reduce_args(c, d_sigma, g, &r);
    hipLaunchKernelGGL(k_sigma_reduce, dim3(r.gx, r.gy), dim3(512), 0, c->stream, r);
    SQD_HIP_CHECK(hipGetLastError());
  }
  return SQD_OK;
}

// ---- batched sigma (sqd_solve_batch).  One operator (mode, spin, ss, shift) applied to one vector of EVERY subspace
// of a batch: the subspaces are grouped by launch class -- kernel instantiation -- and each class goes out as ONE
// launch, blockIdx.z = subspace, sized for the class' largest grid / workgroup / LDS plan.  The arguments never
// change during a Davidson run (the vector is chosen on the device through the state block), so they are written
// once into the batch's staging blob.  Supported classes: the element-gather kernel and the work-item kernel with
// LDS-staged rows in one pass -- every subspace up to a few thousand strings per spin; sigma_batch_supported()
// says so, and callers solve anything else one by one.
bool sigma_batch_supported(const sqd_ctx* c) {
  if (c->sharded()) return false;
  if (c->sig_rows > 0 || c->sig_lists || c->sig_spmm) return false;
  if (c->sig_direct) return true;
  // (R <= 8: the batched launch classes are single-pass, and the single-pass kernel of 9+ columns per thread is the one that
  // does not return on the MI355X -- see launch_sigma_r; such subspaces are solved one by one)
  return c->sig_lds_rows && !(c->sig_ps < c->hv_s.nv_max || c->sig_pd < c->hv_d.nv_max) && c->sig_R <= 8;
}
static int work_item_R(const sqd_ctx* c) {
  const int R = c->sig_R;
  return R <= 1 ? 1 : R <= 2 ? 2 : R <= 4 ? 4 : R <= 8 ? 8 : 16;
}
size_t sigma_batch_bytes(size_t nsub) {
  return nsub * (sizeof(SigmaArgs) + sizeof(DirectArgs) + sizeof(ReduceArgs) + sizeof(DenseArgs) + 4 * 64) + 256;
}
int sigma_batch_plan(const std::vector<sqd_ctx*>& subs, const std::vector<const double*>& d_c,
                     const std::vector<double*>& d_sigma, int mode, bool spin, double ss, double shift,
                     int64_t stride_scale, char* h, char* d, size_t* off_io, SigmaBatchPlan* plan, bool skip_direct) {
  plan->launches.clear();
  const bool sp = (mode == 1 || spin);
  size_t off = *off_io;
  auto take = [&](size_t bytes) {
    off = (off + 63) & ~size_t(63);
    const size_t at = off;
    off += bytes;
    return at;
  };
  const int n = (int)subs.size();
  // class 1: element gather
  {
    std::vector<int> idx;
    for (int p = 0; p < n; ++p)
      if (subs[p]->sig_direct && subs[p]->sig_rows == 0) idx.push_back(p);
    if (!idx.empty() && !skip_direct) {
      const size_t at = take(idx.size() * sizeof(DirectArgs));
      SigmaBatchPlan::Launch L;
      L.kind = 1;
      L.R = 0;
      L.spin = sp;
      L.T = 256;
      L.shmem = 0;
      L.gx = L.gy = 1;
      L.n = (int)idx.size();
      L.args = d + at;
      for (size_t k = 0; k < idx.size(); ++k) {
        sqd_ctx* c = subs[idx[k]];
        DirectArgs g;
        const int64_t st = stride_scale ? c->D : 0;
        fill_direct_args(c, d_c[idx[k]], d_sigma[idx[k]], mode, spin, ss, shift, st, st, &g);
        reinterpret_cast<DirectArgs*>(h + at)[k] = g;
        L.gx = g.gx > L.gx ? g.gx : L.gx;
      }
      plan->launches.push_back(L);
    }
  }
  // the matrix-core same-spin products of the subspaces in dense mode (in front of their work items)
  if (mode == 0) {
    std::vector<int> idx;
    for (int p = 0; p < n; ++p)
      if (!subs[p]->sig_direct && subs[p]->sig_dense) idx.push_back(p);
    if (!idx.empty()) {
      const size_t at = take(idx.size() * sizeof(DenseArgs));
      SigmaBatchPlan::Launch L;
      L.kind = 4;
      L.R = 0;
      L.spin = false;
      L.T = 256;
      L.shmem = 0;
      L.gx = L.gy = 1;
      L.n = (int)idx.size();
      L.args = d + at;
      for (size_t k = 0; k < idx.size(); ++k) {
        sqd_ctx* c = subs[idx[k]];
        DenseArgs da;
        fill_dense_args(c, d_c[idx[k]], stride_scale ? c->D : 0, &da);
        reinterpret_cast<DenseArgs*>(h + at)[k] = da;
        L.gx = da.gx > L.gx ? da.gx : L.gx;
      }
      plan->launches.push_back(L);
    }
  }
  // classes 0: work items, one per template R
  for (int R : {1, 2, 4, 8, 16}) {
    std::vector<int> idx;
    for (int p = 0; p < n; ++p)
      if (!subs[p]->sig_direct && work_item_R(subs[p]) == R) idx.push_back(p);
    if (idx.empty()) continue;
    const size_t at = take(idx.size() * sizeof(SigmaArgs));
    SigmaBatchPlan::Launch L;
    L.kind = 0;
    L.R = R;
    L.spin = sp;
    L.T = 64;
    L.shmem = 0;
    L.gx = L.gy = 1;
    L.n = (int)idx.size();
    L.args = d + at;
    std::vector<ReduceArgs> red;
    for (size_t k = 0; k < idx.size(); ++k) {
      sqd_ctx* c = subs[idx[k]];
      if (!sigma_batch_supported(c)) {
        set_error("sigma_batch_plan: subspace outside the batched launch classes");
        return SQD_ERR_STATE;
      }
      SigmaArgs g;
      const int64_t st = stride_scale ? c->D : 0;
      fill_sigma_args(c, d_c[idx[k]], d_sigma[idx[k]], mode, spin, ss, shift, st, st, &g);
      reinterpret_cast<SigmaArgs*>(h + at)[k] = g;
      L.gx = g.gx > L.gx ? g.gx : L.gx;
      L.gy = g.gy > L.gy ? g.gy : L.gy;
      L.T = g.T > L.T ? g.T : L.T;
      L.shmem = c->sig_shmem > L.shmem ? c->sig_shmem : L.shmem;
      if (sigma_needs_reduce(c, mode, g.vec_index != nullptr)) {
        ReduceArgs r;
        fill_reduce_args(c, d_sigma[idx[k]], g, &r);
        red.push_back(r);
      }
    }
    plan->launches.push_back(L);
    if (!red.empty()) {
      const size_t ar = take(red.size() * sizeof(ReduceArgs));
      SigmaBatchPlan::Launch Lr;
      Lr.kind = 3;
      Lr.R = 0;
      Lr.spin = false;
      Lr.T = 512;
      Lr.shmem = 0;
      Lr.gx = Lr.gy = 1;
      Lr.n = (int)red.size();
      Lr.args = d + ar;
      for (size_t k = 0; k < red.size(); ++k) {
        reinterpret_cast<ReduceArgs*>(h + ar)[k] = red[k];
        Lr.gx = red[k].gx > Lr.gx ? red[k].gx : Lr.gx;
        Lr.gy = red[k].gy > Lr.gy ? red[k].gy : Lr.gy;
      }
      plan->launches.push_back(Lr);
    }
  }
  *off_io = off;
  return SQD_OK;
}

template <int R, bool SPIN>
static int launch_sigma_b(sqd_ctx* c, const SigmaBatchPlan::Launch& L) {
  if (L.shmem > 64 * 1024) {
    static std::atomic<size_t> granted[64];
    const int dev = c->device & 63;
    if (L.shmem > granted[dev].load(std::memory_order_relaxed)) {
      SQD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sigma_b<R, SPIN, true, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)L.shmem));
      granted[dev].store(L.shmem, std::memory_order_relaxed);
    }
  }
  hipLaunchKernelGGL((k_sigma_b<R, SPIN, true, false>), dim3(L.gx, L.gy, (unsigned)L.n), dim3(L.T), L.shmem, c->stream,
                     reinterpret_cast<const SigmaArgs*>(L.args));
  return SQD_OK;
}
int sigma_batch_launch(sqd_ctx* parent, const SigmaBatchPlan& plan) {
  for (const SigmaBatchPlan::Launch& L : plan.launches) {
    if (L.kind == 1) {
      if (L.spin)
        hipLaunchKernelGGL((k_sigma_direct_b<true>), dim3(L.gx, 1, (unsigned)L.n), dim3(256), 0, parent->stream,
                           reinterpret_cast<const DirectArgs*>(L.args));
      else
        hipLaunchKernelGGL((k_sigma_direct_b<false>), dim3(L.gx, 1, (unsigned)L.n), dim3(256), 0, parent->stream,
                           reinterpret_cast<const DirectArgs*>(L.args));
    } else if (L.kind == 4) {
      hipLaunchKernelGGL(k_same_spin_mfma_b, dim3(L.gx, DENSE_SPLIT, (unsigned)L.n), dim3(256), 0, parent->stream,
                         reinterpret_cast<const DenseArgs*>(L.args));
    } else if (L.kind == 3) {
      hipLaunchKernelGGL(k_sigma_reduce_b, dim3(L.gx, L.gy, (unsigned)L.n), dim3(512), 0, parent->stream,
                         reinterpret_cast<const ReduceArgs*>(L.args));
    } else {
      int rc = SQD_OK;
      switch (L.R) {
        case 1: rc = L.spin ? launch_sigma_b<1, true>(parent, L) : launch_sigma_b<1, false>(parent, L); break;
        case 2: rc = L.spin ? launch_sigma_b<2, true>(parent, L) : launch_sigma_b<2, false>(parent, L); break;
        case 4: rc = L.spin ? launch_sigma_b<4, true>(parent, L) : launch_sigma_b<4, false>(parent, L); break;
        case 8: rc = L.spin ? launch_sigma_b<8, true>(parent, L) : launch_sigma_b<8, false>(parent, L); break;
        default: rc = L.spin ? launch_sigma_b<16, true>(parent, L) : launch_sigma_b<16, false>(parent, L); break;
      }
      if (rc != SQD_OK) return rc;
    }
    SQD_HIP_CHECK(hipGetLastError());
  }
  return SQD_OK;
}

// benchmark hook (sqd_time_dense): `reps` launches of the matrix-core same-spin product of the current subspace on
// `copies` identical argument records at once -- copies = 1 is the launch of a single solve (25 tiles x DENSE_SPLIT
// workgroups at 317 x 317: latency, not throughput), copies = 16 fills the chip the way a batched solve of 16 subspaces
// does (every copy reads the same operands and writes the same partial products: timing only)
int time_dense_product(sqd_ctx* c, const double* d_c, int reps, int copies, double* ms, double* flops) {
  if (!c->sig_dense || c->sig_spmm) {
    set_error("the current subspace does not use the dense same-spin product");
    return SQD_ERR_STATE;
  }
  if (copies < 1) copies = 1;
  if (copies > 64) copies = 64;
  DenseArgs d;
  fill_dense_args(c, d_c, 0, &d);
  d.stop = nullptr;
  d.vec_index = nullptr;
  std::vector<DenseArgs> h((size_t)copies, d);
  SQD_TRY(c->io_out.reserve((size_t)copies * sizeof(DenseArgs)));
  SQD_HIP_CHECK(hipMemcpyAsync(c->io_out.p, h.data(), (size_t)copies * sizeof(DenseArgs), hipMemcpyHostToDevice, c->stream));
  SQD_STREAM_SYNC(c->stream);
  auto launch = [&]() {
    hipLaunchKernelGGL(k_same_spin_mfma_b, dim3(d.gx, DENSE_SPLIT, (unsigned)copies), dim3(256), 0, c->stream,
                       static_cast<const DenseArgs*>(c->io_out.p));
  };
  launch();
  SQD_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
  for (int i = 0; i < reps; ++i) launch();
  SQD_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
  SQD_HIP_CHECK(hipGetLastError());
  SQD_STREAM_SYNC(c->stream);
  float t = 0.f;
  SQD_HIP_CHECK(hipEventElapsedTime(&t, c->ev[2], c->ev[3]));
  *ms = (double)t / reps;
  // G = H_a C + C H_b on the padded orders: 2 pa^2 pb + 2 pa pb^2 flops per copy
  const double pa = c->dense_pa, pb = c->dense_pb;
  *flops = (double)copies * (2.0 * pa * pa * pb + 2.0 * pa * pb * pb);
  return SQD_OK;
}

int apply_h(sqd_ctx* c, const double* d_c, double* d_sigma, int use_spin, double ss, double shift, int64_t in_stride,
            int64_t out_stride) {
  if (use_spin == 3) {
    // pyscf SpinPenaltyFCISolver.contract_2e: sz = |neleca - nelecb| / 2 decides between the two penalty forms
    const double sz = 0.5 * std::abs(c->nelec[0] - c->nelec[1]);
    use_spin = (ss < sz * (sz + 1.0) + 0.1) ? 1 : 2;
  }
  if (use_spin == 0) return launch_sigma(c, d_c, d_sigma, 0, false, 0.0, 0.0, in_stride, out_stride);
  if (use_spin == 1) return launch_sigma(c, d_c, d_sigma, 0, true, ss, shift, in_stride, out_stride);
  if (use_spin != 2) {
    set_error("use_spin must be 0..3");
    return SQD_ERR_INVALID;
  }
  // sigma = H c + shift * (S^2 - ss)^2 c     (pyscf fix_spin_, second form)
  const int64_t D = c->D;
  SQD_TRY(c->tmp1.reserve(D * 8));
  SQD_TRY(c->tmp2.reserve(D * 8));
  double* t1 = c->tmp1.as<double>();
  double* t2 = c->tmp2.as<double>();
  const unsigned nb_ = (unsigned)((D + 255) / 256 > 2048 ? 2048 : (D + 255) / 256);
  const int* stop = c->sigma_stop;
  const int* idx = (c->sigma_index && (in_stride || out_stride)) ? c->sigma_index : nullptr;
  SQD_TRY(launch_sigma(c, d_c, t1, 1, false, 0.0, 0.0, in_stride, 0));                          // t1 = S^2 c
  hipLaunchKernelGGL(k_axpby, dim3(nb_), dim3(256), 0, c->stream, D, -ss, d_c, 1.0, t1, stop, idx, in_stride,
                     (int64_t)0);                                                                  // t1 -= ss c
  SQD_TRY(launch_sigma(c, t1, t2, 1, false, 0.0, 0.0, 0, 0));                                    // t2 = S^2 t1
  hipLaunchKernelGGL(k_axpby, dim3(nb_), dim3(256), 0, c->stream, D, -ss, (const double*)t1, 1.0, t2, stop,
                     (const int*)nullptr, (int64_t)0, (int64_t)0);                                 // t2 -= ss t1
  SQD_TRY(launch_sigma(c, d_c, d_sigma, 0, false, 0.0, 0.0, in_stride, out_stride));             // sigma = H c
  hipLaunchKernelGGL(k_axpby, dim3(nb_), dim3(256), 0, c->stream, D, shift, (const double*)t2, 1.0, d_sigma, stop, idx,
                     (int64_t)0, out_stride);
  SQD_HIP_CHECK(hipGetLastError());
  return SQD_OK;
}

}  // namespace sqd

#ifdef SQD_PHASE_CLOCK
// out[16]: the column sums over the workgroups' rows ([0..7] own-row items, [8..15] alpha-single batch items)
extern "C" __attribute__((visibility("default"))) int sqd_probe_clk_sigma(unsigned long long* out, int reset) {
  static std::vector<unsigned long long> h((size_t)sqd::SCLK_ROWS * sqd::SCLK_COLS);
  if (out) {
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(sqd::sqd_clk_sigma), h.size() * 8) != hipSuccess) return -1;
    for (int k = 0; k < sqd::SCLK_COLS; ++k) out[k] = 0;
    for (size_t r = 0; r < (size_t)sqd::SCLK_ROWS; ++r)
      for (int k = 0; k < sqd::SCLK_COLS; ++k) out[k] += h[r * sqd::SCLK_COLS + k];
  }
  if (reset) {
    std::fill(h.begin(), h.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(sqd::sqd_clk_sigma), h.data(), h.size() * 8) != hipSuccess) return -1;
  }
  return 0;
}
#endif
