// sigma = P H P c for LARGE string sets with short, even link lists -- uniform-random sets from a few thousand strings
// per spin up: BASELINE config 2 read literally (10^4 x 10^4 strings, D = 10^8, 800 MB per vector) and the "subspace
// dimensions of ~10^7" of the reference's README (README.md:78).  Replaces, like the other sigma kernels, pyscf
// selected_ci.contract_2e + contract_ss behind kernel_fixed_space (reference qiskit_addon_sqd/fermion.py:721-723,
// :810-818).  Three launches per sigma, each element of C leaving HBM once per side:
//
//   * BETA side, a list pass on C (k_sigma_lists): the merged same-spin list of a beta string (singles' one-body value,
//     then doubles: ~11 links at 10^4 strings) lives IN REGISTERS of the lane that owns the string: a workgroup of 1024
//     lanes owns a block of <= 1024 columns, loads their lists ONCE and then walks a chunk of ~400 rows of the matrix.
//     Every row is staged in LDS (80 KB at 10^4 columns; the next row is prefetched into registers while this one is
//     evaluated), so a link costs one LDS gather and one FMA: no pointer -> record -> operand chain, no record traffic per
//     row at all.  Columns of a block are sorted by list length (wavefronts then run uniform trip counts without
//     padding); the result row is un-permuted through LDS so that global loads and stores stay coalesced.  Lists longer
//     than the register file's share (16 links / 4 single links) keep their tail in a small LDS table.  The diagonal
//     term is formed in the (natural-order, coalesced) epilogue from hdiag, whose lines -- like the row's own -- are
//     pulled into the L2 two rows ahead (the epilogue's operands are HBM misses otherwise, and the row loop of a
//     workgroup cannot run faster than the latency of what it asks for and uses inside one iteration).  The ten
//     column-block workgroups that read the same rows are placed on ONE XCD (block b runs on XCD b % 8): nine of ten
//     row reads are L2 hits.
//   * ALPHA side by rows, on C itself (k_alpha_rows): sigma[A][.] += sum over the list of A of value x C[A'][.] -- whole
//     row segments streamed with 16 bytes per lane, panel by panel (see the kernel).  It adds onto the beta pass's
//     result and brings the compact term with it, so the beta pass's row loop has no operand from HBM but hdiag.
//     (Round 4's first half ran the alpha side as a second list pass on C^T between two transpositions: 1.59 + 0.40 ms
//     against 0.8 ms; removed.)
//     (The other order -- k_alpha_rows writes, the list pass adds onto it in its epilogue, the row of sigma prefetched
//     into the L2 like hdiag's -- moves the 800 MB read from a bandwidth-bound kernel into this latency-bound loop:
//     alpha rows 1.00 -> 0.88 ms, list pass 0.99 -> 1.09 ms; 2.074 -> 2.056 ms, not kept: profiles/r04b/order_swap_probe.txt.)
//   * The terms that pair a single link of each spin (2.7 % of the links, but a nested loop per element in the row
//     kernel) are evaluated for the strings that HAVE single links (~2600 x 2600 at 10^4 x 10^4) by a small kernel of
//     their own (k_lists_t4; operands gathered from C in place, the lists from per-string records).
// Fixed order of accumulation everywhere: the same bits on every run.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <numeric>

#include "sqd_common.h"

namespace sqd {

namespace {

constexpr int LT = 1024;        // column slots of a workgroup
#ifndef SQD_LISTS_CPL
#define SQD_LISTS_CPL 1
#endif
#ifndef SQD_LISTS_REGCAP
#define SQD_LISTS_REGCAP 16
#endif
#ifndef SQD_LISTS_ROT
#define SQD_LISTS_ROT 0
#endif
#ifndef SQD_LISTS_L2PF
#define SQD_LISTS_L2PF 2
#endif
#ifndef SQD_LISTS_SPREAD
#define SQD_LISTS_SPREAD 1
#endif
constexpr int CPL = SQD_LISTS_CPL;  // columns per lane: the fixed per-lane state (addresses, masks, the next row's share) is paid
                                // once per CPL columns, which is what lets 24 links per column stay in registers
constexpr int NT = LT / CPL;    // lanes of a workgroup
constexpr int REGCAP = SQD_LISTS_REGCAP;  // same-spin links of a column held in registers
constexpr int SCAP = 4;         // single links of a column held in registers (opposite-spin terms)
constexpr int OVL_CAP = 512;    // per block: same-spin links beyond REGCAP (LDS)
constexpr int OVS_CAP = 256;    // per block: single links beyond SCAP (LDS)
constexpr int NPF2 = 5 * CPL;   // 16-byte pieces of the next row a lane prefetches (rows longer than 2 NPF2 NT: second phase)
constexpr int RPC_MAX = 512;    // rows of a row chunk at most (their per-row scalars sit in LDS)

typedef double lists_d2 __attribute__((ext_vector_type(2)));
__device__ inline int l_ctz64(uint64_t x) { return __ffsll((long long)x) - 1; }
__device__ inline int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }
// element `i` (a lane's 32-bit index, i * 8 < 2^32) of an array whose base is wave-uniform: written so that the compiler
// sees base + zext(32-bit BYTE offset) and emits the scalar-base form of global_load / global_store (one VGPR of offset);
// indexed the usual way the offset is a 64-bit quantity per access -- ISA: a zero high dword kept in a VGPR for each of
// the twenty loads of the next row's share
__device__ inline double ldu(const double* base, unsigned i) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + (i << 3));
}
__device__ inline double2 ldu2(const double* base, unsigned i) {  // 16-byte element i; base must be 16-byte aligned
  return *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(base) + (i << 4));
}
__device__ inline void stu(double* base, unsigned i, double v) {
  *reinterpret_cast<double*>(reinterpret_cast<char*>(base) + (i << 3)) = v;
}

// ---------------------------------------------------------------------------------------------------------------
// tables
// ---------------------------------------------------------------------------------------------------------------
struct ListFillArgs {
  int nblk;
  GPtr<const int32_t> col;        // [nblk * LT] string address of the slot, -1 = padding
  GPtr<const uint32_t> desc;      // [nblk * LT * 2]: {links | singles << 16, overflow starts: links | singles << 16}
  GPtr<const int64_t> s_ptr, d_ptr;
  GPtr<const SRec> s_rec;
  GPtr<const double> s_val;
  GPtr<const uint32_t> d_src;
  GPtr<const double> d_val;
  GPtr<uint32_t> ridx;            // [nblk][REGCAP / 2][LT]: two 16-bit source addresses per word
  GPtr<double> rval;              // [nblk][REGCAP][LT]
  GPtr<uint32_t> sing;            // [nblk][SCAP][LT]: src | widx << 16 | sign << 31
  GPtr<uint32_t> ovl_idx;         // [nblk][OVL_CAP]
  GPtr<double> ovl_val;           // [nblk][OVL_CAP]
  GPtr<uint32_t> ovs;             // [nblk][OVS_CAP]
};
__device__ inline uint32_t pack_single(const SRec r) {
  return (r.src & 0xffffu) | (srec_widx(r.meta) << 16) | (r.meta & 0x80000000u);
}
// One thread per column slot.  (A conflict-aware ORDER of the register-held links -- the lanes of an LDS service group
// choosing in turn, round by round, the remaining link whose bank pair is least taken -- cut the list pass's conflict
// cycles by 38 % and its LDS-active cycles by 17 % (profiles/r04b/alpha_rows_probe_5.txt), and its time by 0.7 %: the
// pass is not bound by LDS throughput; the ordering cost 0.6 ms of table build.  Removed again.)
__global__ void __launch_bounds__(LT) k_lists_fill(const ListFillArgs g) {
  const int b = blockIdx.x;
  const int64_t slot = (int64_t)b * LT + threadIdx.x;
  const int col = g.col[slot];
  int64_t s0 = 0, s1 = 0, d0 = 0, d1 = 0;
  if (col >= 0) {
    s0 = g.s_ptr[col];
    s1 = g.s_ptr[col + 1];
    d0 = g.d_ptr[col];
    d1 = g.d_ptr[col + 1];
  }
  const int ns = (int)(s1 - s0), nl = ns + (int)(d1 - d0);
  const uint32_t ov = g.desc[2 * slot + 1];
  const int ovl0 = (int)(ov & 0xffffu), ovs0 = (int)(ov >> 16);
  uint32_t lsrc[REGCAP];
  double lval[REGCAP];
  for (int k = 0; k < (nl > REGCAP ? nl : REGCAP); ++k) {
    uint32_t src = 0;
    double val = 0.0;
    if (k < nl) {
      if (k < ns) {
        src = g.s_rec[s0 + k].src;
        val = g.s_val[s0 + k];
      } else {
        src = g.d_src[d0 + (k - ns)];
        val = g.d_val[d0 + (k - ns)];
      }
    }
    if (k < REGCAP) {
      lsrc[k] = src;
      lval[k] = val;
    } else {
      g.ovl_idx[(int64_t)b * OVL_CAP + ovl0 + (k - REGCAP)] = src;
      g.ovl_val[(int64_t)b * OVL_CAP + ovl0 + (k - REGCAP)] = val;
    }
  }
  for (int k = 0; k < REGCAP; k += 2) {
    g.rval[((int64_t)b * REGCAP + k) * LT + threadIdx.x] = lval[k];
    g.rval[((int64_t)b * REGCAP + k + 1) * LT + threadIdx.x] = lval[k + 1];
    g.ridx[((int64_t)b * (REGCAP / 2) + (k >> 1)) * LT + threadIdx.x] = (lsrc[k] & 0xffffu) | (lsrc[k + 1] << 16);
  }
  for (int j = 0; j < (ns > SCAP ? ns : SCAP); ++j) {
    const uint32_t w = (j < ns) ? pack_single(g.s_rec[s0 + j]) : 0u;
    if (j < SCAP) g.sing[((int64_t)b * SCAP + j) * LT + threadIdx.x] = w;
    else g.ovs[(int64_t)b * OVS_CAP + ovs0 + (j - SCAP)] = w;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// single x single (and the S^2 exchange term), for the strings that HAVE single links (compact lists clist_a / clist_b):
//   T[ia][ib] = sum_{alpha singles (A <- A', wa, s)} s sum_{beta singles (B <- B', wb, t)} t W(wa, wb) C[A'][B'],
//   W = (pq|rs)  (+ pen when the beta link undoes the alpha link's orbital move); direct_element's last loop
// One launch, operands gathered from C itself.  Until the end of round 4 this was two: k_lists_compact gathered the
// compact matrix C[clist_a][clist_b] (56 us at 10^4 x 10^4: per thread ten dependent pairs of round trips) and a
// per-element kernel walked string -> CSR pointers -> records -> compact addresses -> operand for every element (a
// 64-bit division and six dependent round trips each: 103 us); both were bound by the LENGTH of their chains, not by
// bytes (the compact row staged in LDS changed nothing: 88 us).  Now the chains are walked once per subspace
// (k_lists_t4_tab: per compact row / column the first two single links as one 16-byte record) and an element costs two
// round trips: its records, then its operands -- C[A'][B'] read in place (the 8-byte gathers of a workgroup fall into the
// 1250 lines of one or two rows of C, every line asked for 2.5 times at short distance: L2 hits after the first).
// Order of accumulation unchanged (beta links inside alpha links): the same bits.
// ---------------------------------------------------------------------------------------------------------------
struct ListT4Args {
  int64_t ma, mb, nb, c_stride;
  GPtr<const double> c;
  GPtr<const uint32_t> clist_a, clist_b;
  GPtr<const int64_t> sa_ptr, sb_ptr;
  GPtr<const SRec> sa_rec, sb_rec;
  GPtr<const double> eri_pp;
  int nnorb, mode, spin;
  double pen;
  GPtr<const uint4> tab_a;  // [ma]: {meta 0, source string 0, meta 1, source string 1}
  GPtr<const uint4> tab_b;  // [mb]: {meta 0, meta 1, source string 0 | source string 1 << 16, number of single links}
  GPtr<const int32_t> cnt_a;  // [ma]: number of single links
  GPtr<double> t4;
  GPtr<const int> stop, vec_index;
};
struct ListT4TabArgs {
  int64_t ma, mb;
  GPtr<const uint32_t> clist_a, clist_b;
  GPtr<const int64_t> sa_ptr, sb_ptr;
  GPtr<const SRec> sa_rec, sb_rec;
  GPtr<uint4> tab_a, tab_b;
  GPtr<int32_t> cnt_a;
};
__global__ void __launch_bounds__(256) k_lists_t4_tab(const ListT4TabArgs g) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < g.ma) {
    const int64_t A = g.clist_a[i];
    const int64_t s0 = g.sa_ptr[A];
    const int n = (int)(g.sa_ptr[A + 1] - s0);
    uint4 t = make_uint4(0u, 0u, 0u, 0u);
    if (n > 0) {
      const SRec r = g.sa_rec[s0];
      t.x = r.meta;
      t.y = r.src;
    }
    if (n > 1) {
      const SRec r = g.sa_rec[s0 + 1];
      t.z = r.meta;
      t.w = r.src;
    }
    g.tab_a[i] = t;
    g.cnt_a[i] = n;
  }
  if (i < g.mb) {
    const int64_t B = g.clist_b[i];
    const int64_t s0 = g.sb_ptr[B];
    const int n = (int)(g.sb_ptr[B + 1] - s0);
    uint4 t = make_uint4(0u, 0u, 0u, (uint32_t)n);
    if (n > 0) {
      const SRec r = g.sb_rec[s0];
      t.x = r.meta;
      t.z = r.src & 0xffffu;  // (nb <= 65535 on this path: lists_select)
    }
    if (n > 1) {
      const SRec r = g.sb_rec[s0 + 1];
      t.y = r.meta;
      t.z |= r.src << 16;
    }
    g.tab_b[i] = t;
  }
}
constexpr int T4_CPT = 4;
#ifndef SQD_T4_T
#define SQD_T4_T 256
#endif
constexpr int T4_T = SQD_T4_T;
// workgroup (x, y): compact row x, compact columns y * 4 T4_T ..., four per thread (a whole row per workgroup up to 4096
// columns: the 8-byte gathers of a row meet again in the L2 -- with 1024 columns per workgroup every workgroup pulled
// 60 % of the row's lines for itself: 93 us)
__global__ void __launch_bounds__(T4_T) k_lists_t4(const ListT4Args g) {
  HIP_DYNAMIC_SHARED(double, w_l)  // [nnorb]
  if (g.stop && *g.stop) return;
  const int64_t vsel = g.vec_index ? (int64_t)(*g.vec_index - 1) : 0;
  const double* __restrict__ C = g.c + vsel * g.c_stride;
  const int64_t ia = blockIdx.x;
  const uint4 ta = g.tab_a[ia];
  const int na_l = g.cnt_a[ia];
  const int mb = (int)g.mb;
  uint4 tb[T4_CPT];
  double acc[T4_CPT];
#pragma unroll
  for (int u = 0; u < T4_CPT; ++u) {
    const int ib = (int)blockIdx.y * (T4_T * T4_CPT) + u * T4_T + (int)threadIdx.x;
    tb[u] = (ib < mb) ? g.tab_b[ib] : make_uint4(0u, 0u, 0u, 0u);
    acc[u] = 0.0;
  }
  for (int la = 0; la < na_l; ++la) {
    uint32_t meta_a, src_a;
    if (la == 0) {
      meta_a = ta.x;
      src_a = ta.y;
    } else if (la == 1) {
      meta_a = ta.z;
      src_a = ta.w;
    } else {  // longer lists: from the CSR tables
      const SRec ra = g.sa_rec[g.sa_ptr[g.clist_a[ia]] + la];
      meta_a = ra.meta;
      src_a = ra.src;
    }
    const double* __restrict__ srow = C + (int64_t)src_a * g.nb;
    const double* __restrict__ w = g.eri_pp + (int64_t)(srec_widx(meta_a) >> 1) * g.nnorb;
    const int partner = (int)srec_widx(meta_a) ^ 1;
    const double sgn_a = srec_sign(meta_a);
    // the alpha link's row of (pq|rs) through LDS: gathered from memory it cost as much as the operands themselves --
    // a 64-lane gather occupies the CU's address path for one clock per distinct line whether the line is in the L1 or not
    if (la) __syncthreads();
    for (int i = threadIdx.x; i < g.nnorb; i += T4_T) w_l[i] = w[i];
    __syncthreads();
    // every load unconditional (a link that does not exist reads element 0 and is not added): behind a branch each
    // pair of loads was waited for before the next branch was even looked at -- eight round trips in sequence per alpha link
    double wv[T4_CPT][2], cv[T4_CPT][2];
#pragma unroll
    for (int u = 0; u < T4_CPT; ++u)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const bool on = k < (int)tb[u].w;
        const uint32_t meta = k ? tb[u].y : tb[u].x;
        const uint32_t col = on ? (k ? (tb[u].z >> 16) : (tb[u].z & 0xffffu)) : 0u;
        wv[u][k] = w_l[on ? (srec_widx(meta) >> 1) : 0u];
        cv[u][k] = srow[col];
      }
#pragma unroll
    for (int u = 0; u < T4_CPT; ++u) {
      const int n = (int)tb[u].w;
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const uint32_t meta = k ? tb[u].y : tb[u].x;
        double x = (g.mode == 0) ? wv[u][k] : 0.0;
        if (g.spin) x += ((int)srec_widx(meta) == partner) ? g.pen : 0.0;
        const double t1 = t + srec_sign(meta) * x * cv[u][k];
        t = (k < n) ? t1 : t;
      }
      if (n > 2) {
        const int ib = (int)blockIdx.y * (T4_T * T4_CPT) + u * T4_T + (int)threadIdx.x;
        const int64_t sb0 = g.sb_ptr[g.clist_b[ib]];
        for (int k = 2; k < n; ++k) {
          const SRec rb = g.sb_rec[sb0 + k];
          double x = (g.mode == 0) ? w[srec_widx(rb.meta) >> 1] : 0.0;
          if (g.spin) x += ((int)srec_widx(rb.meta) == partner) ? g.pen : 0.0;
          t += srec_sign(rb.meta) * x * srow[rb.src];
        }
      }
      acc[u] += sgn_a * t;
    }
  }
#pragma unroll
  for (int u = 0; u < T4_CPT; ++u) {
    const int ib = (int)blockIdx.y * (T4_T * T4_CPT) + u * T4_T + (int)threadIdx.x;
    if (ib < mb) g.t4[ia * g.mb + ib] = acc[u];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// the alpha side by rows, on C itself (round 4, second half): sigma_a[A][.] = sum over the merged same-spin list of
// A of value x C[A'][.]  (+ sign J_beta[.][pair] C[A'][.] for the single links) -- rows of 80 KB streamed with 16 bytes
// per lane, no LDS, no transposition either way.  Every element is read once per link of its row (~11 x 800 MB at
// 10^4 x 10^4), which is what round 3's k_sigma_rows did beside everything else; here it is a pass of its own and the
// tasks are ordered PANEL-major: all rows of a panel of columns (sized for the 256 MB Infinity Cache) before the next
// panel, so that the eleven reads of a row segment find it on the die and HBM sees every element once.
// A wavefront owns (row, panel): its lists sit one link per lane and are broadcast with readlane (scalar base address +
// 32-bit lane offset per request), eight requests in flight per lane.  Fixed order: the same bits on every run.
// Measured at 10^4 x 10^4: 0.88 ms, 10.9 TB/s into the CUs whatever the number of requests in flight per lane (4 .. 12);
// 19 TB/s at 3000 x 3000 where a panel fits the L2s.  A quarter of a wavefront per row with 32-column panels private to
// an XCD's L2 (lists broadcast inside the 16-lane groups with ds_bpermute) ran 0.92 ms alone, 2.11 against 2.14 ms in the
// whole sigma; with the lists read per lane from memory 1.2 ms (profiles/r04b/alpha_rows_probe_*.txt).  Not kept.
// ---------------------------------------------------------------------------------------------------------------
struct AlphaRowsArgs {
  GPtr<const double> in;
  GPtr<double> out;
  int64_t in_stride, out_stride, na, nb;
  int pw, npanel;                 // panel width in columns (a multiple of 128), panels
  int chunk;                      // links of a list held by the lanes at a time: 64 (test hook SQD_ALPHA_CHUNK: fewer)
  GPtr<const int64_t> s_ptr, d_ptr;
  GPtr<const SRec> s_rec;
  GPtr<const double> s_val;
  GPtr<const uint32_t> d_src;
  GPtr<const double> d_val;
  GPtr<const double> jT;          // [nnorb][nb]
  int accum;                      // out already holds the beta pass's part of the same elements: add onto it
  GPtr<const double> t4;          // compact single x single term [ma][mb], null: none
  int64_t t4_ld;
  GPtr<const int32_t> cidx_a, cidx_b;
  GPtr<const int> stop, vec_index;
};
#ifndef SQD_AR_K
#define SQD_AR_K 8
#endif
#ifndef SQD_AR_WAVES
#define SQD_AR_WAVES 7  // (8 wavefronts per SIMD spill two registers: 1.01 against 0.88 ms at 10^4 x 10^4)
#endif
constexpr int AR_K = SQD_AR_K;    // requests in flight per lane
template <bool WIDE>
struct ArPair {
  // the two columns of a lane inside a 128-column segment: WIDE (16-byte requests) 2 lane, 2 lane + 1; else lane, lane + 64
  static __device__ inline double2 ld(const double* seg, unsigned lane, int left) {
    if (WIDE) return (int)(2 * lane) < left ? ldu2(seg, lane) : make_double2(0.0, 0.0);
    double2 v = make_double2(0.0, 0.0);
    if ((int)lane < left) v.x = ldu(seg, lane);
    if ((int)lane + 64 < left) v.y = ldu(seg, lane + 64);
    return v;
  }
  static __device__ inline void st(double* seg, unsigned lane, int left, double2 v) {
    if (WIDE) {
      if ((int)(2 * lane) < left) *reinterpret_cast<double2*>(reinterpret_cast<char*>(seg) + (lane << 4)) = v;
    } else {
      if ((int)lane < left) stu(seg, lane, v.x);
      if ((int)lane + 64 < left) stu(seg, lane + 64, v.y);
    }
  }
};
__device__ inline double readlane_f64(double v, int l) {
  uint32_t w[2];
  __builtin_memcpy(w, &v, 8);
  w[0] = (uint32_t)__builtin_amdgcn_readlane((int)w[0], l);
  w[1] = (uint32_t)__builtin_amdgcn_readlane((int)w[1], l);
  double r;
  __builtin_memcpy(&r, w, 8);
  return r;
}
template <bool WIDE>
__global__ void __launch_bounds__(256, SQD_AR_WAVES) k_alpha_rows(const AlphaRowsArgs g) {
  if (g.stop && *g.stop) return;
  const int64_t vsel = g.vec_index ? (int64_t)(*g.vec_index - 1) : 0;
  const double* __restrict__ C = g.in + vsel * g.in_stride;
  double* __restrict__ out = g.out + vsel * g.out_stride;
  const unsigned lane = threadIdx.x & 63u;
  const int64_t task = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int p = uniform_int((int)(task / g.na));
  if (p >= g.npanel) return;
  const int64_t A = (int64_t)uniform_int((int)(task - (int64_t)p * g.na));
  const int64_t nb = g.nb;
  const int64_t s0 = g.s_ptr[A], d0 = g.d_ptr[A];
  const int ns = uniform_int((int)(g.s_ptr[A + 1] - s0)), nd = uniform_int((int)(g.d_ptr[A + 1] - d0));
  const int c_lo = p * g.pw, c_hi = (int)((int64_t)c_lo + g.pw < nb ? (int64_t)c_lo + g.pw : nb);
  const int crow = g.t4 ? uniform_int(g.cidx_a[A]) : -1;  // compact row of A (strings with single links), -1: none
  // the lists, one link per lane (lists longer than 64: reloaded per segment, 64 at a time)
  const int CH = g.chunk;
  uint32_t v_ssrc = 0, v_smeta = 0, v_dsrc = 0;
  double v_sval = 0.0, v_dval = 0.0;
  auto load_s = [&](int base) {
    if ((int)lane < CH && base + (int)lane < ns) {
      const SRec r = g.s_rec[s0 + base + lane];
      v_ssrc = r.src;
      v_smeta = r.meta;
      v_sval = g.s_val[s0 + base + lane];
    }
  };
  auto load_d = [&](int base) {
    if ((int)lane < CH && base + (int)lane < nd) {
      v_dsrc = g.d_src[d0 + base + lane];
      v_dval = g.d_val[d0 + base + lane];
    }
  };
  if (ns > 0 && ns <= CH) load_s(0);
  if (nd > 0 && nd <= CH) load_d(0);
  for (int cs = c_lo; cs < c_hi; cs += 128) {
    const int left = c_hi - cs;  // columns of this segment (and beyond) that exist
    double2 acc = make_double2(0.0, 0.0);
    if (g.accum) acc = ArPair<WIDE>::ld(out + A * nb + cs, lane, left);
    if (crow >= 0) {  // (uniform) single x single, from the compact matrix
      const int ca = WIDE ? (int)(2 * lane) : (int)lane, cb2 = WIDE ? ca + 1 : ca + 64;
      const double* trow = g.t4 + (int64_t)crow * g.t4_ld;
      if (ca < left) {
        const int ci = g.cidx_b[cs + ca];
        if (ci >= 0) acc.x += trow[ci];
      }
      if (cb2 < left) {
        const int ci = g.cidx_b[cs + cb2];
        if (ci >= 0) acc.y += trow[ci];
      }
    }
    // single links: one-body value + sign x J_beta[column][pair]
    for (int base = 0; base < ns; base += CH) {
      if (ns > CH) load_s(base);
      const int m = ns - base < CH ? ns - base : CH;
      for (int l = 0; l < m; ++l) {
        const uint32_t src = __builtin_amdgcn_readlane((int)v_ssrc, l), meta = __builtin_amdgcn_readlane((int)v_smeta, l);
        const double val = readlane_f64(v_sval, l);
        const double2 x = ArPair<WIDE>::ld(C + (int64_t)src * nb + cs, lane, left);
        const double2 j = ArPair<WIDE>::ld(g.jT + (int64_t)(srec_widx(meta) >> 1) * nb + cs, lane, left);
        const double sgn = (meta >> 31) ? -1.0 : 1.0;
        acc.x += (val + sgn * j.x) * x.x;
        acc.y += (val + sgn * j.y) * x.y;
      }
    }
    // double links, AR_K requests in flight
    for (int base = 0; base < nd; base += CH) {
      if (nd > CH) load_d(base);
      const int m = nd - base < CH ? nd - base : CH;
      for (int l = 0; l < m; l += AR_K) {
        double2 x[AR_K];
#pragma unroll
        for (int k = 0; k < AR_K; ++k) {
          x[k] = make_double2(0.0, 0.0);
          if (l + k < m) {  // (uniform)
            const uint32_t src = __builtin_amdgcn_readlane((int)v_dsrc, l + k);
            x[k] = ArPair<WIDE>::ld(C + (int64_t)src * nb + cs, lane, left);
          }
        }
#pragma unroll
        for (int k = 0; k < AR_K; ++k) {
          if (l + k < m) {  // (the values are broadcast where they are used: no registers held across the requests)
            const double val = readlane_f64(v_dval, l + k);
            acc.x += val * x[k].x;
            acc.y += val * x[k].y;
          }
        }
      }
    }
    ArPair<WIDE>::st(out + A * nb + cs, lane, left, acc);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// the list pass
// ---------------------------------------------------------------------------------------------------------------
struct ListsArgs {
  GPtr<const double> in;      // C [n_r][n_c]: the matrix whose rows are staged
  GPtr<double> out;           // [n_r][n_c]
  int64_t in_stride, out_stride;  // added per selected vector (vec_index)
  int64_t n_r, n_c;
  int mode, spin;
  double ss, shift, szterm;
  int nblk, cpb, cpx, rpc;    // column blocks, columns per block, row chunks per XCD, rows per chunk
  int norb, nnorb;
  int pitch, o_jr, o_ob, o_ovlv, o_ovli, o_ovs, o_rs;  // LDS plan, in doubles
  GPtr<const int32_t> col;
  GPtr<const uint32_t> desc, ridx, sing, ovl_idx, ovs;
  GPtr<const double> rval, ovl_val;
  GPtr<const int32_t> wlen;   // [nblk][16][4]: per wavefront {list trips, list tail, single trips, single tail}
  GPtr<const uint64_t> strs_c, strs_r;
  GPtr<const double> hdiag, jrow;
  GPtr<const int32_t> cidx_c, cidx_r;
  GPtr<const double> t4;      // compact single x single term [m_r][m_c] (pure S^2 operator: no alpha pass to bring it), null: none
  int64_t t4_ld;
  GPtr<const int> stop, vec_index;
  int dbg;  // tuning hook (SQD_LISTS_DBG): bit 0 plain block order instead of the XCD-aware one, 1 no gathers, 2 no staging of the next row
};

// The packed source addresses of a lane's links are loop invariants, and left alone the compiler unpacks them ONCE in
// front of the row loop -- into one VGPR per link instead of one per two links (ISA: 24 v_lshl_add_u32 results kept
// live), which is what pushed the kernel over its 128 registers.  Passing the word through an empty asm statement
// makes it opaque, so the two ALU operations of the unpacking stay inside the loop.
__device__ inline uint32_t opaque(uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
  asm volatile("" : "+v"(w));
#endif
  return w;
}
// a value that was loaded only to pull its line into the L2: "used" here so that the load is neither dropped nor waited
// for anywhere else
__device__ inline void consume(uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
  asm volatile("" ::"v"(w));
#else
  (void)w;
#endif
}

// ---- phase clocks (probe builds only: -DSQD_PHASE_CLOCK; profiles/probes/_lists_clock.py): thread 0 of every workgroup
// adds the 100 MHz wall-clock deltas of the phases of its row loop into its own row of a device array
#ifdef SQD_PHASE_CLOCK
constexpr int LCLK_ROWS = 1024, LCLK_COLS = 8;
__device__ unsigned long long sqd_clk_lists[4 * LCLK_ROWS * LCLK_COLS];
#define LCLK_MARK(slot)                                                                  \
  do {                                                                                   \
    const unsigned long long t_ = wall_clock64();                                        \
    if (threadIdx.x == 0) sqd_clk_lists[(VAR * LCLK_ROWS + blockIdx.x % LCLK_ROWS) * LCLK_COLS + (slot)] += t_ - lclk_t; \
    lclk_t = t_;                                                                         \
  } while (0)
#else
#define LCLK_MARK(slot)
#endif
// VAR 1: H; 2: H + shift (S^2 - ss); 3: the pure S^2 operator (diagonal + compact term only)
template <int VAR>
__global__ void __launch_bounds__(NT) k_sigma_lists(const ListsArgs g) {
  constexpr bool SPIN = VAR >= 2, lists = VAR != 3, HMODE = VAR != 3, T4 = VAR == 3;
  HIP_DYNAMIC_SHARED(double, smem)
  if (g.stop && *g.stop) return;
  const int64_t vsel = g.vec_index ? (int64_t)(*g.vec_index - 1) : 0;
  const double* __restrict__ M = g.in + vsel * g.in_stride;
  double* __restrict__ out = g.out + vsel * g.out_stride;
  // block -> (row chunk, column block): the column blocks of one row chunk share an XCD (block b runs on XCD b % 8).
  // (10^4 columns: 3 chunks x 10 column blocks fill 30 of an XCD's 32 CUs.  Giving the two idle CUs a short last chunk of
  // the XCD's rows, its column blocks one after the other, was measured both ways: as a loop over column blocks inside
  // the workgroup the pass gains 5.5 % on its rows but the loop costs the kernel its registers (119 -> 128 VGPRs, 43
  // spilled SGPRs: +6 % per row); as ten more workgroups at the end of the grid, which start as CUs fall free, it is
  // slower than without them -- 0.994 -> 1.034 ms at 48 rows: the late ones run behind the full chunks
  // (profiles/r04b/extra_rows_probe.txt).  Not kept.)
  unsigned xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
  if (g.dbg & 1) xcd = 0, q = blockIdx.x;
  const int cb = (int)(q % (unsigned)g.nblk);
  const int64_t chunk = (int64_t)xcd * g.cpx + q / (unsigned)g.nblk;
  const int64_t r0 = chunk * g.rpc;
  const int64_t r1 = (r0 + g.rpc < g.n_r) ? r0 + g.rpc : g.n_r;
  if (r0 >= r1) return;
  const int n_c = (int)g.n_c;
  const int tid = threadIdx.x;
  double* __restrict__ jr = smem + g.o_jr;
  double* __restrict__ ob = smem + g.o_ob;
  double* __restrict__ ovlv = smem + g.o_ovlv;
  uint32_t* __restrict__ ovli = reinterpret_cast<uint32_t*>(smem + g.o_ovli);
  uint32_t* __restrict__ ovsl = reinterpret_cast<uint32_t*>(smem + g.o_ovs);
  // per-row scalars of this workgroup's row chunk: the alpha strings, their compact indices
  uint64_t* __restrict__ rs_str = reinterpret_cast<uint64_t*>(smem + g.o_rs);
  int* __restrict__ rs_cid = reinterpret_cast<int*>(smem + g.o_rs + g.rpc);

  // ---- this lane's CPL columns and their lists, once.  Slot s = c * NT + tid of the block's LT slots.
  const int c0 = cb * g.cpb;                          // first column of the block (natural order)
  const int ncol = (n_c - c0) < g.cpb ? (n_c - c0) : g.cpb;
  int col[CPL];                                       // (sorted by list length inside the block; -1 = padding)
  uint32_t d0[CPL], d1[CPL];                          // {links | singles << 16}, overflow starts
  int trips[CPL], wtail_l[CPL], strips[CPL], wtail_s[CPL];  // wave-uniform trip counts
  uint32_t ri[CPL][REGCAP / 2];
  double rv[CPL][REGCAP];
  uint32_t sg[CPL][SCAP];
  uint64_t sN[CPL];  // string of the NATURAL-order column c0 + c * NT + tid (epilogue: spin penalty diagonal)
  int cidN[CPL];     // its compact index (single x single term)
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int64_t slot = (int64_t)cb * LT + c * NT + tid;
    col[c] = g.col[slot];
    d0[c] = g.desc[2 * slot];
    d1[c] = g.desc[2 * slot + 1];
    const int32_t* wl = g.wlen + ((int64_t)cb * (LT / 64) + c * (NT / 64) + (tid >> 6)) * 4;
    trips[c] = (g.dbg & 2) ? 0 : uniform_int(wl[0]);
    wtail_l[c] = uniform_int(wl[1]);
    strips[c] = uniform_int(wl[2]);
    wtail_s[c] = uniform_int(wl[3]);
    if (lists) {
#pragma unroll
      for (int k = 0; k < REGCAP / 2; ++k) ri[c][k] = g.ridx[((int64_t)cb * (REGCAP / 2) + k) * LT + c * NT + tid];
#pragma unroll
      for (int k = 0; k < REGCAP; ++k) rv[c][k] = g.rval[((int64_t)cb * REGCAP + k) * LT + c * NT + tid];
#pragma unroll
      for (int j = 0; j < SCAP; ++j) sg[c][j] = g.sing[((int64_t)cb * SCAP + j) * LT + c * NT + tid];
    } else {
#pragma unroll
      for (int k = 0; k < REGCAP / 2; ++k) ri[c][k] = 0;
#pragma unroll
      for (int k = 0; k < REGCAP; ++k) rv[c][k] = 0.0;
#pragma unroll
      for (int j = 0; j < SCAP; ++j) sg[c][j] = 0;
    }
    sN[c] = 0;
    cidN[c] = -1;
    if (c * NT + tid < ncol) {
      if (SPIN) sN[c] = g.strs_c[c0 + c * NT + tid];
      if (T4 && g.t4) cidN[c] = g.cidx_c[c0 + c * NT + tid];
    }
  }
  if (lists) {
    for (int i = tid; i < OVL_CAP; i += NT) {
      ovlv[i] = g.ovl_val[(int64_t)cb * OVL_CAP + i];
      ovli[i] = g.ovl_idx[(int64_t)cb * OVL_CAP + i];
    }
    for (int i = tid; i < OVS_CAP; i += NT) ovsl[i] = g.ovs[(int64_t)cb * OVS_CAP + i];
  }

  // ---- rows travel as their 16-byte ALIGNED image: a row that starts on an odd double is fetched from the double in
  // front of it (the previous row's last element, or the neighbouring vector's: always inside the allocation) and
  // LDS holds that image; row = img + (0 or 1).  16 bytes per lane and request: half the vector-memory instructions
  // of 8-byte loads for the same bytes.
  double* __restrict__ img = smem;
  const double* __restrict__ row;
  {
    const double* rp = M + r0 * g.n_c;
    const unsigned sh = (unsigned)((reinterpret_cast<uintptr_t>(rp) >> 3) & 1u);
    const unsigned n2 = ((unsigned)n_c + sh + 1u) >> 1;
    for (unsigned b = tid; b < n2; b += NT) reinterpret_cast<double2*>(img)[b] = ldu2(rp - sh, b);
    row = img + sh;
  }
  if (lists)
    for (int i = tid; i < g.nnorb; i += NT) jr[i] = g.jrow[r0 * g.nnorb + i];
  for (int i = tid; i < (int)(r1 - r0); i += NT) {
    rs_str[i] = g.strs_r[r0 + i];
    rs_cid[i] = (T4 && g.t4) ? g.cidx_r[r0 + i] : -1;
  }
  // Every load so far -- the lists above all -- is waited for HERE: left pending into the loop, the compiler's wait
  // counts inside it are sized for the first iteration (s_waitcnt vmcnt(0..4) in front of the first uses of the list
  // registers in the first gather round: ISA), and in every later iteration such a count waits for the requests that
  // iteration has just issued -- the next row's first piece, hdiag, the previous row's store.
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
  __syncthreads();

#ifdef SQD_PHASE_CLOCK
  unsigned long long lclk_t = wall_clock64();
#endif
  // The column-block workgroups of a row chunk run in step and ask for the same 16 KB piece of the same row at the same
  // moment; with SQD_LISTS_ROT piece u of block cb is piece (u + cb) mod NPF2 of the row, so that the blocks are spread
  // over the row (and the L2 channels it is interleaved over) at any one time.
#if SQD_LISTS_ROT
  const int rot = uniform_int(cb % NPF2);
#define SQD_PU(u) (((u) + rot) >= NPF2 ? ((u) + rot - NPF2) : ((u) + rot))
#else
#define SQD_PU(u) (u)
#endif
  // L2 prefetch, SQD_LISTS_L2PF rows ahead: what an iteration asks for and uses -- the hdiag segment of the epilogue,
  // the next row for whichever of the column blocks comes first -- is an HBM miss otherwise, and the loop's period
  // cannot be shorter than the latency of that.  The ~126 lines (this block's hdiag segment, this block's tenth of the
  // row of C) are dealt to the sixteen wavefronts, eight lines each: lanes 0..7 take one line each, the other lanes
  // repeat lane 7's.  The values are never used: they are "consumed" one iteration later, where the requests issued
  // behind them have long landed.
  // EVERY load of the loop is unconditional (indices clamped, not predicated): a load under a lane predicate is a
  // branch around it, the compiler then no longer knows how many loads follow the one it must wait for, and waits for
  // all of them -- s_waitcnt vmcnt(0) in the epilogue: for the prefetch too, an HBM access per row (ISA).
#if SQD_LISTS_L2PF
  const char* pfb;  // this lane's line in row r0 (advances by a row per iteration)
  {
    const unsigned row_lines = (unsigned)((n_c * 8 + 127) / 128), lpb = (row_lines + (unsigned)g.nblk - 1u) / (unsigned)g.nblk;
    const unsigned hd_lines = HMODE ? (unsigned)((ncol * 8 + 127) / 128) : 0u;
    unsigned my_lines = lpb;  // lines of the row's share that exist
    if ((unsigned)cb * lpb >= row_lines) my_lines = 0;
    else if ((unsigned)cb * lpb + lpb > row_lines) my_lines = row_lines - (unsigned)cb * lpb;
    const unsigned total = hd_lines + my_lines;  // (>= 1: the block has columns)
    const unsigned lane = (unsigned)tid & 63u, wv = (unsigned)tid >> 6;
    unsigned L = wv * 8u + (lane < 8u ? lane : 7u);
    if (L >= total) L = total - 1u;
    pfb = (L < hd_lines) ? reinterpret_cast<const char*>(g.hdiag + r0 * n_c + c0) + ((size_t)L << 7)
                         : reinterpret_cast<const char*>(M + r0 * n_c) + ((size_t)((unsigned)cb * lpb + (L - hd_lines)) << 7);
  }
  uint32_t pfx = 0;
#define SQD_LISTS_PREFETCH()                                                                                   \
  do {                                                                                                         \
    consume(pfx);                                                                                              \
    const int64_t ahead = (r + SQD_LISTS_L2PF < r1 ? r + SQD_LISTS_L2PF : r1 - 1) - r0;                        \
    pfx = *reinterpret_cast<const uint32_t*>(pfb + ahead * (int64_t)n_c * 8);                                  \
  } while (0)
#else
#define SQD_LISTS_PREFETCH()
#endif
  for (int64_t r = r0; r < r1; ++r) {
    // -- 1. requests: the next row (it moves into LDS behind the barrier) and what the epilogue of THIS row adds.
    // Nothing here may be USED before the barrier: a use is a wait for every request issued before it.
    const bool more = r + 1 < r1 && !(g.dbg & 4);
    double2 pf[NPF2];
    double pj = 0.0;
    double t4v[CPL], hd[CPL], own[CPL];
    // (the last row of a chunk "asks" for itself again: the loads stay unconditional, the result is not stored)
    const double* nrow = M + (more ? r + 1 : r) * g.n_c;  // (uniform base + 32-bit lane offset: saddr loads)
    const unsigned sh_n = (unsigned)((reinterpret_cast<uintptr_t>(nrow) >> 3) & 1u);
    const unsigned n2_n = ((unsigned)n_c + sh_n + 1u) >> 1;
    const double* __restrict__ nimg = nrow - sh_n;
#define SQD_LISTS_REQUEST(u)                                                                    \
  do {                                                                                          \
    const unsigned i_ = (unsigned)(SQD_PU(u) * NT) + (unsigned)tid;                             \
    pf[u] = ldu2(nimg, i_ < n2_n ? i_ : n2_n - 1u);                                             \
  } while (0)
    // SQD_LISTS_SPREAD: one piece up front, one behind each round of gathers (the address pipe works in the shadow of the
    // LDS round trips) instead of two up front and three behind the first round
    constexpr int UPF = SQD_LISTS_SPREAD ? (NPF2 > REGCAP / 4 ? NPF2 - REGCAP / 4 : 0) : NPF2 / 2;
#pragma unroll
    for (int u = 0; u < UPF; ++u) SQD_LISTS_REQUEST(u);
    if (lists) pj = ldu(g.jrow + (more ? r + 1 : r) * g.nnorb, (unsigned)(tid < g.nnorb ? tid : g.nnorb - 1));
    const int crow = T4 ? rs_cid[r - r0] : -1;  // (per-row scalars of the chunk come from LDS: a scalar load from
                                                // memory here would be a ~1 us wait for the whole workgroup per row)
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      t4v[c] = 0.0;
      hd[c] = 0.0;
      own[c] = 0.0;
      // the diagonal term is formed in the epilogue, in natural column order: hdiag is read coalesced (the bit loop
      // over a table of the row in LDS that round 4's first version used cost eight LDS reads per element) and the
      // element itself comes from the staged row
      const int lcq = c * NT + tid < ncol ? c * NT + tid : ncol - 1;
      if (HMODE) hd[c] = ldu(g.hdiag + r * n_c + c0, (unsigned)lcq);
      own[c] = row[c0 + lcq];
      if (T4 && cidN[c] >= 0 && crow >= 0) t4v[c] = ldu(g.t4 + (int64_t)crow * g.t4_ld, (unsigned)cidN[c]);  // single x single (compact)
    }
    LCLK_MARK(0);
    // -- 2. this row from LDS
    double acc[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      acc[c] = 0.0;
    }
    if (lists) {
      // four links per round.  The single links come FIRST in a merged list and SCAP = 4 of them have their packed
      // {source, pair, sign} in registers: their second term -- sign x J_alpha[row][pair] x the same operand -- is added in
      // the first round, from the operand that round has just gathered (one LDS read for J instead of two per link)
      static_assert(SCAP == 4, "the first gather round carries the single links' J term");
#pragma unroll
      for (int k0 = 0; k0 < REGCAP; k0 += 4) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
          if (k0 < trips[c]) {
            const uint32_t wa = opaque(ri[c][k0 / 2]), wb = opaque(ri[c][k0 / 2 + 1]);
            const double x0 = row[wa & 0xffffu], x1 = row[wa >> 16], x2 = row[wb & 0xffffu], x3 = row[wb >> 16];
            if (k0 == 0) {
              const int ns = (int)(d0[c] >> 16);
              const double xs[SCAP] = {x0, x1, x2, x3};
#pragma unroll
              for (int j = 0; j < SCAP; ++j) {
                if (j < strips[c]) {
                  const uint32_t w = opaque(sg[c][j]);
                  const double sgn = (w >> 31) ? -1.0 : 1.0;
                  const double coef = (j < ns) ? sgn * jr[(w >> 17) & 0xfffu] : 0.0;
                  acc[c] += coef * xs[j];
                }
              }
            }
            acc[c] += rv[c][k0] * x0;
            acc[c] += rv[c][k0 + 1] * x1;
            acc[c] += rv[c][k0 + 2] * x2;
            acc[c] += rv[c][k0 + 3] * x3;
          }
        }
        if (SQD_LISTS_SPREAD) {
          if (UPF + k0 / 4 < NPF2) SQD_LISTS_REQUEST(UPF + k0 / 4);
          if (k0 + 4 >= REGCAP) SQD_LISTS_PREFETCH();
        } else if (k0 == 0) {  // the second half of the next row's requests, behind the first round of gathers
#pragma unroll
          for (int u = UPF; u < NPF2; ++u) SQD_LISTS_REQUEST(u);
          SQD_LISTS_PREFETCH();
        }
      }
    } else {
#pragma unroll
      for (int u = UPF; u < NPF2; ++u) SQD_LISTS_REQUEST(u);
      SQD_LISTS_PREFETCH();
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      if (lists) {
        if (wtail_l[c] > 0) {
          const int nl = (int)(d0[c] & 0xffffu), ovl0 = (int)(d1[c] & 0xffffu);
          for (int t = 0; t < wtail_l[c]; ++t) {
            const bool on = REGCAP + t < nl;
            const int e = on ? ovl0 + t : 0;
            const double v = on ? ovlv[e] : 0.0;
            acc[c] += v * row[ovli[e]];
          }
        }
        const int ns = (int)(d0[c] >> 16);
        if (wtail_s[c] > 0) {
          const int ovs0 = (int)(d1[c] >> 16);
          for (int t = 0; t < wtail_s[c]; ++t) {
            const bool on = SCAP + t < ns;
            const uint32_t w = ovsl[on ? ovs0 + t : 0];
            const double sgn = (w >> 31) ? -1.0 : 1.0;
            const double coef = on ? sgn * jr[(w >> 17) & 0xfffu] : 0.0;
            acc[c] += coef * row[w & 0xffffu];
          }
        }
      }
      // -- 3. back to natural column order through LDS
      if (col[c] >= 0) ob[col[c] - c0] = acc[c];
    }
    LCLK_MARK(1);
    __syncthreads();
    LCLK_MARK(2);
    // -- 4. the finished row moves out (coalesced)
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int lc = c * NT + tid;  // natural-order column of the block
      if (lc < ncol) {
        double v = ob[lc] + t4v[c];
        double d = hd[c];
        if (SPIN) {
          const double pop = (double)__popcll(sN[c] & ~rs_str[r - r0]);  // beta occupied, alpha empty
          d = HMODE ? d + g.shift * (g.szterm + pop - g.ss) : g.szterm + pop;
        }
        v += d * own[c];
        stu(out + r * n_c + c0, (unsigned)lc, v);
      }
    }
    LCLK_MARK(3);
    // -- 5. the next row moves in -- behind the epilogue: its requests have had the epilogue's time as well to land
    if (more) {
#pragma unroll
      for (int u = 0; u < NPF2; ++u)
        if ((unsigned)(SQD_PU(u) * NT) < n2_n && (unsigned)tid < n2_n - (unsigned)(SQD_PU(u) * NT))
          reinterpret_cast<double2*>(img)[tid + SQD_PU(u) * NT] = pf[u];
      for (unsigned b = tid + NPF2 * NT; b < n2_n; b += NT)  // (rows beyond 2 NPF2 NT columns)
        reinterpret_cast<double2*>(img)[b] = ldu2(M + (r + 1) * g.n_c - sh_n, b);
      row = img + sh_n;
      if (lists && tid < g.nnorb) jr[tid] = pj;
      if (lists)
        for (int i = tid + NT; i < g.nnorb; i += NT) jr[i] = g.jrow[(r + 1) * g.nnorb + i];  // (nnorb > NT: norb > 31 at CPL 2)
    }
    LCLK_MARK(4);
    __syncthreads();
    LCLK_MARK(5);
  }
#if SQD_LISTS_L2PF
  consume(pfx);
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// host: plan
// ---------------------------------------------------------------------------------------------------------------
struct SidePlan {
  int nblk = 0, cpb = 0;
  std::vector<int32_t> col, wlen;
  std::vector<uint32_t> desc;
  std::vector<int32_t> cidx;
  std::vector<uint32_t> clist;
  bool ok = true;
};

// columns of each block sorted by list length (longest first), 64 at a time dealt to the wavefronts so that the four
// SIMDs of a CU carry the same load (a workgroup's wavefronts go to the SIMDs in cyclic order)
void plan_side(const int64_t* s_ptr, const int64_t* d_ptr, int64_t n, SidePlan& p, bool compact_only) {
  p.nblk = (int)((n + LT - 1) / LT);
  p.cpb = (int)((n + p.nblk - 1) / p.nblk);
  p.col.assign((size_t)p.nblk * LT, -1);
  p.desc.assign((size_t)p.nblk * LT * 2, 0u);
  p.wlen.assign((size_t)p.nblk * 16 * 4, 0);
  p.cidx.assign((size_t)n, -1);
  p.clist.clear();
  p.ok = true;
  for (int64_t I = 0; I < n; ++I)
    if (s_ptr[I + 1] > s_ptr[I]) {
      p.cidx[(size_t)I] = (int32_t)p.clist.size();
      p.clist.push_back((uint32_t)I);
    }
  if (compact_only) {  // (alpha side: k_alpha_rows reads the CSR lists as they are)
    p.col.clear();
    p.desc.clear();
    p.wlen.clear();
    return;
  }
  std::vector<int> order;
  for (int b = 0; b < p.nblk; ++b) {
    const int64_t c0 = (int64_t)b * p.cpb, c1 = std::min<int64_t>(n, c0 + p.cpb);
    const int m = (int)(c1 - c0);
    order.resize((size_t)m);
    std::iota(order.begin(), order.end(), 0);
    auto len = [&](int i) { return (s_ptr[c0 + i + 1] - s_ptr[c0 + i]) + (d_ptr[c0 + i + 1] - d_ptr[c0 + i]); };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return len(x) > len(y); });
    int ovl = 0, ovs = 0;
    for (int rank = 0; rank < m; ++rank) {
      // group j of 64 sorted columns -> (wavefront, column index c of its lanes), slot = c * NT + wave * 64 + lane.
      // CPL 2: groups j and 15 - j share a wavefront (equal sums); CPL 1: a snake over the four SIMDs of the CU
      const int j = rank / 64, lane = rank % 64;
      int wave, cc;
      if (CPL == 2) {
        wave = j < 8 ? j : 15 - j;
        cc = j < 8 ? 0 : 1;
      } else {
        const int grp = (j % 8 < 4) ? j % 4 : 3 - (j % 4);
        wave = grp + 4 * (j / 4);
        cc = 0;
      }
      const int slot = cc * NT + wave * 64 + lane;
      const int64_t I = c0 + order[(size_t)rank];
      const int ns = (int)(s_ptr[I + 1] - s_ptr[I]), nl = ns + (int)(d_ptr[I + 1] - d_ptr[I]);
      if (nl > 60000 || ns > 60000) p.ok = false;
      const size_t s = (size_t)b * LT + slot;
      p.col[s] = (int32_t)I;
      p.desc[2 * s] = (uint32_t)nl | ((uint32_t)ns << 16);
      p.desc[2 * s + 1] = (uint32_t)ovl | ((uint32_t)ovs << 16);
      int32_t* wl = &p.wlen[((size_t)b * 16 + slot / 64) * 4];
      const int trips = std::min(REGCAP, (nl + 3) / 4 * 4);
      wl[0] = std::max(wl[0], trips);
      wl[1] = std::max(wl[1], nl > REGCAP ? nl - REGCAP : 0);
      wl[2] = std::max(wl[2], std::min(SCAP, ns));
      wl[3] = std::max(wl[3], ns > SCAP ? ns - SCAP : 0);
      if (nl > REGCAP) ovl += nl - REGCAP;
      if (ns > SCAP) ovs += ns - SCAP;
    }
    if (ovl > OVL_CAP || ovs > OVS_CAP) p.ok = false;
  }
}

struct LdsPlan {
  int pitch, o_jr, o_ob, o_ovlv, o_ovli, o_ovs, o_rs;
  size_t bytes;
};
// LDS of the pass whose staged rows have n_c doubles
bool lds_plan(int64_t n_c, int nnorb, int lds_bytes, LdsPlan& L) {
  int off = (int)((n_c + 3) & ~int64_t(1));  // the aligned image of a row: up to one double in front, one behind
  L.pitch = off;
  L.o_jr = off, off += (nnorb + 1) & ~1;
  L.o_ob = off, off += LT;
  L.o_ovlv = off, off += OVL_CAP;
  L.o_ovli = off, off += OVL_CAP / 2;
  L.o_ovs = off, off += OVS_CAP / 2;
  L.o_rs = off, off += (3 * RPC_MAX + 1) / 2;  // per-row scalars of a chunk (8 + 4 bytes a row)
  L.bytes = (size_t)off * 8;
  return L.bytes <= (size_t)lds_bytes;
}

}  // namespace

// state of the list path, owned by the context (sqd_common.h: sqd_ctx::lists)
struct ListSideDev {
  int nblk = 0, cpb = 0;
  int64_t m = 0;  // strings with single links
  DevBuf col, desc, wlen, ridx, rval, sing, ovl_idx, ovl_val, ovs, cidx, clist;  // (alpha side: cidx and clist only)
};
struct ListState {
  ListSideDev side[2];
  SidePlan plan[2];  // host copies (their uploads are asynchronous)
  DevBuf t4, t4tab_a, t4tab_b, t4cnt_a;
  LdsPlan lds_b;     // the list pass (rows of C: nb doubles)
};

void lists_release(sqd_ctx* c) {
  if (!c->lists) return;
  ListState* s = static_cast<ListState*>(c->lists);
  for (auto& sd : s->side)
    for (DevBuf* b : {&sd.col, &sd.desc, &sd.wlen, &sd.ridx, &sd.rval, &sd.sing, &sd.ovl_idx, &sd.ovl_val, &sd.ovs,
                      &sd.cidx, &sd.clist})
      b->release();
  s->t4.release();
  s->t4tab_a.release();
  s->t4tab_b.release();
  s->t4cnt_a.release();
  delete s;
  c->lists = nullptr;
}

// Is the list path possible / chosen for the subspace whose CSR pointers are on the host?  (phase 2 of set_subspace)
bool lists_select(sqd_ctx* c, int64_t na, int64_t nb, int64_t row0, int64_t row1, const int64_t* tot, const int* nocc) {
  c->sig_lists = false;
  const char* env = std::getenv("SQD_SIGMA_LISTS");
  const int forced = env ? std::atoi(env) : -1;
  if (forced == 0) return false;
  if (row0 != 0 || row1 != na) return false;                  // whole-subspace contexts only
  if (nb > 65535) return false;                               // 16-bit source addresses in the beta pass's registers
  if (nocc[0] < 1 || nocc[1] < 1) return false;
  if (!c->lists) c->lists = new ListState();
  ListState* s = static_cast<ListState*>(c->lists);
  if (!lds_plan(nb, c->nnorb, c->lds_bytes, s->lds_b)) return false;
  if (forced != 1) {
    // Lists short and even enough for the registers to hold nearly all of them, sets large enough for the table build
    // (a host sort of the columns + one fill launch: +0.8 ms of set_subspace) to pay within a Davidson solve.  Measured on
    // the MI355X (profiles/r04b/alpha_rows_probe_5.txt, uniform N x N, ms per sigma, this path | k_sigma_rows):
    // 10 000: 2.14 | 3.82; 7 000: 0.96 | 1.46; 5 000: 0.43 | 0.52; 4 000: 0.25 | 0.29; 3 000: 0.133 | 0.150;
    // 2 000: 0.067 | 0.062.
    if (na < 5000 || nb < 5000) return false;
    if (tot[0] + tot[1] > 16 * na || tot[2] + tot[3] > 16 * nb) return false;
    if (tot[0] > 2 * na || tot[2] > 2 * nb) return false;
  }
  plan_side(c->h_sptr, c->h_dptr, na, s->plan[0], true);
  plan_side(c->h_sptr_b, c->h_dptr_b, nb, s->plan[1], false);
  if (!s->plan[0].ok || !s->plan[1].ok) return false;
  c->sig_lists = true;
  return true;
}

// columns of a panel of k_alpha_rows: na rows x pw columns within the share of the Infinity Cache that holds a panel
static int alpha_panel_width(int64_t na, int64_t nb) {
  static const double mb = [] {
    const char* env = std::getenv("SQD_ALPHA_PANEL_MB");
    return env ? std::atof(env) : 16.0;
  }();
  int64_t pw = (int64_t)(mb * 1048576.0 / (8.0 * (double)na)) / 128 * 128;
  const int64_t full = (nb + 127) / 128 * 128;
  if (pw < 128) pw = 128;
  if (pw > full) pw = full;
  return (int)pw;
}

template <class T>
static int upload_vec(sqd_ctx* c, DevBuf& buf, const std::vector<T>& v) {
  SQD_TRY(buf.reserve(std::max<size_t>(v.size() * sizeof(T), 16)));
  if (!v.empty()) SQD_HIP_CHECK(hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
  return SQD_OK;
}

// device tables of the list path; enqueued behind launch C of set_subspace (the CSR lists must be filled)
int lists_build(sqd_ctx* c) {
  ListState* s = static_cast<ListState*>(c->lists);
  for (int sp = 0; sp < 2; ++sp) {
    const SidePlan& p = s->plan[sp];
    ListSideDev& d = s->side[sp];
    const SpinTables& t = c->sp[sp];
    d.nblk = p.nblk;
    d.cpb = p.cpb;
    d.m = (int64_t)p.clist.size();
    SQD_TRY(upload_vec(c, d.cidx, p.cidx));
    SQD_TRY(upload_vec(c, d.clist, p.clist));
    if (sp == 0) continue;  // (alpha side: k_alpha_rows reads the CSR lists as they are)
    SQD_TRY(upload_vec(c, d.col, p.col));
    SQD_TRY(upload_vec(c, d.desc, p.desc));
    SQD_TRY(upload_vec(c, d.wlen, p.wlen));
    SQD_TRY(d.ridx.reserve((size_t)p.nblk * (REGCAP / 2) * LT * 4));
    SQD_TRY(d.rval.reserve((size_t)p.nblk * REGCAP * LT * 8));
    SQD_TRY(d.sing.reserve((size_t)p.nblk * SCAP * LT * 4));
    SQD_TRY(d.ovl_idx.reserve((size_t)p.nblk * OVL_CAP * 4));
    SQD_TRY(d.ovl_val.reserve((size_t)p.nblk * OVL_CAP * 8));
    SQD_TRY(d.ovs.reserve((size_t)p.nblk * OVS_CAP * 4));
    SQD_HIP_CHECK(hipMemsetAsync(d.ovl_idx.p, 0, (size_t)p.nblk * OVL_CAP * 4, c->stream));
    SQD_HIP_CHECK(hipMemsetAsync(d.ovl_val.p, 0, (size_t)p.nblk * OVL_CAP * 8, c->stream));
    SQD_HIP_CHECK(hipMemsetAsync(d.ovs.p, 0, (size_t)p.nblk * OVS_CAP * 4, c->stream));
    ListFillArgs f;
    f.nblk = p.nblk;
    f.col = d.col.as<int32_t>();
    f.desc = d.desc.as<uint32_t>();
    f.s_ptr = t.s_ptr.as<int64_t>();
    f.d_ptr = t.d_ptr.as<int64_t>();
    f.s_rec = t.s_rec.as<SRec>();
    f.s_val = t.s_val.as<double>();
    f.d_src = t.d_src.as<uint32_t>();
    f.d_val = t.d_val.as<double>();
    f.ridx = d.ridx.as<uint32_t>();
    f.rval = d.rval.as<double>();
    f.sing = d.sing.as<uint32_t>();
    f.ovl_idx = d.ovl_idx.as<uint32_t>();
    f.ovl_val = d.ovl_val.as<double>();
    f.ovs = d.ovs.as<uint32_t>();
    hipLaunchKernelGGL(k_lists_fill, dim3(p.nblk), dim3(LT), 0, c->stream, f);
    SQD_HIP_CHECK(hipGetLastError());
  }
  const int64_t ma = s->side[0].m, mb = s->side[1].m;
  SQD_TRY(s->t4.reserve(std::max<size_t>((size_t)ma * mb * 8, 16)));
  SQD_TRY(s->t4tab_a.reserve(std::max<size_t>((size_t)ma * 16, 16)));
  SQD_TRY(s->t4tab_b.reserve(std::max<size_t>((size_t)mb * 16, 16)));
  SQD_TRY(s->t4cnt_a.reserve(std::max<size_t>((size_t)ma * 4, 16)));
  if (ma > 0 && mb > 0) {
    ListT4TabArgs f;
    f.ma = ma;
    f.mb = mb;
    f.clist_a = s->side[0].clist.as<uint32_t>();
    f.clist_b = s->side[1].clist.as<uint32_t>();
    f.sa_ptr = c->sp[0].s_ptr.as<int64_t>();
    f.sb_ptr = c->sp[1].s_ptr.as<int64_t>();
    f.sa_rec = c->sp[0].s_rec.as<SRec>();
    f.sb_rec = c->sp[1].s_rec.as<SRec>();
    f.tab_a = s->t4tab_a.as<uint4>();
    f.tab_b = s->t4tab_b.as<uint4>();
    f.cnt_a = s->t4cnt_a.as<int32_t>();
    hipLaunchKernelGGL(k_lists_t4_tab, dim3((unsigned)((std::max(ma, mb) + 255) / 256)), dim3(256), 0, c->stream, f);
    SQD_HIP_CHECK(hipGetLastError());
  }
  return SQD_OK;
}

// arguments of the list pass: beta lists on C (rows = alpha strings)
static void fill_pass_args(sqd_ctx* c, ListState* s, int mode, bool spin, double ss, double shift, ListsArgs* gp) {
  ListsArgs& g = *gp;
  std::memset(&g, 0, sizeof(g));
  const ListSideDev& d = s->side[1];
  const LdsPlan& L = s->lds_b;
  g.n_r = c->na;
  g.n_c = c->nb;
  g.mode = mode;
  g.spin = spin ? 1 : 0;
  g.ss = ss;
  g.shift = shift;
  const double sz = 0.5 * (c->nelec[0] - c->nelec[1]);
  g.szterm = sz * (sz + 1.0);
  g.nblk = d.nblk;
  g.cpb = d.cpb;
  // row chunks per XCD: the XCD's 32 CUs over the column blocks, but no chunk shorter than 8 rows
  g.cpx = (int)std::min<int64_t>(std::max(1, 32 / d.nblk), std::max<int64_t>(1, (g.n_r + 63) / 64));
  g.cpx = (int)std::max<int64_t>(g.cpx, (g.n_r + 8 * (RPC_MAX - 8) - 1) / (8 * (RPC_MAX - 8)));  // chunks of <= RPC_MAX rows
  const int64_t nchunks = 8 * (int64_t)g.cpx;
  int64_t rpc = (g.n_r + nchunks - 1) / nchunks;
  rpc = (rpc + 7) / 8 * 8;
  g.rpc = (int)rpc;
  g.norb = c->norb;
  g.nnorb = c->nnorb;
  g.pitch = L.pitch;
  g.o_jr = L.o_jr;
  g.o_ob = L.o_ob;
  g.o_ovlv = L.o_ovlv;
  g.o_ovli = L.o_ovli;
  g.o_ovs = L.o_ovs;
  g.o_rs = L.o_rs;
  g.col = d.col.as<int32_t>();
  g.desc = d.desc.as<uint32_t>();
  g.ridx = d.ridx.as<uint32_t>();
  g.sing = d.sing.as<uint32_t>();
  g.ovl_idx = d.ovl_idx.as<uint32_t>();
  g.ovs = d.ovs.as<uint32_t>();
  g.rval = d.rval.as<double>();
  g.ovl_val = d.ovl_val.as<double>();
  g.wlen = d.wlen.as<int32_t>();
  g.strs_c = c->sp[1].strs.as<uint64_t>();
  g.strs_r = c->sp[0].strs.as<uint64_t>();
  g.hdiag = c->hdiag.as<double>();
  g.jrow = c->sp[0].jrow.as<double>();  // J of the staged rows' spin
  g.cidx_c = d.cidx.as<int32_t>();
  g.cidx_r = s->side[0].cidx.as<int32_t>();
  g.stop = c->sigma_stop;
  static const int dbg = [] {
    const char* env = std::getenv("SQD_LISTS_DBG");
    return env ? std::atoi(env) : 0;
  }();
  g.dbg = dbg;
}

int launch_sigma_lists(sqd_ctx* c, const double* d_c, double* d_sigma, int mode, bool spin, double ss, double shift,
                       int64_t in_stride, int64_t out_stride) {
  ListState* s = static_cast<ListState*>(c->lists);
  const bool sp = (mode == 1 || spin);
  const bool indexed = c->sigma_index && (in_stride || out_stride);
  const int* vec_index = indexed ? c->sigma_index : nullptr;
  const int64_t na = c->na, nb = c->nb, ma = s->side[0].m, mb = s->side[1].m;
  const bool cross = ma > 0 && mb > 0;     // single x single term exists
  const bool alpha_pass = (mode == 0);     // the pure S^2 operator has no same-spin part
  static const int pass_mask = [] {        // profiling hook: bit 1 single x single term, 2 alpha pass, 3 beta pass (bit 0: unused)
    const char* env = std::getenv("SQD_LISTS_PASSES");
    return env ? std::atoi(env) : 15;
  }();
  // launch 1: the single x single term of the strings that have single links
  if (cross && (pass_mask & 2)) {
    ListT4Args t;
    t.ma = ma;
    t.mb = mb;
    t.nb = nb;
    t.c_stride = in_stride;
    t.c = d_c;
    t.clist_a = s->side[0].clist.as<uint32_t>();
    t.clist_b = s->side[1].clist.as<uint32_t>();
    t.sa_ptr = c->sp[0].s_ptr.as<int64_t>();
    t.sb_ptr = c->sp[1].s_ptr.as<int64_t>();
    t.sa_rec = c->sp[0].s_rec.as<SRec>();
    t.sb_rec = c->sp[1].s_rec.as<SRec>();
    t.eri_pp = c->eri_pp.as<double>();
    t.nnorb = c->nnorb;
    t.mode = mode;
    t.spin = sp ? 1 : 0;
    t.pen = (mode == 1) ? -1.0 : -shift;
    t.tab_a = s->t4tab_a.as<uint4>();
    t.tab_b = s->t4tab_b.as<uint4>();
    t.cnt_a = s->t4cnt_a.as<int32_t>();
    t.t4 = s->t4.as<double>();
    t.stop = c->sigma_stop;
    t.vec_index = vec_index;
    hipLaunchKernelGGL(k_lists_t4, dim3((unsigned)ma, (unsigned)((mb + T4_T * T4_CPT - 1) / (T4_T * T4_CPT))), dim3(T4_T),
                       (size_t)c->nnorb * 8, c->stream, t);
    SQD_HIP_CHECK(hipGetLastError());
  }
  // launch 3, the list pass: diagonal + beta lists on C (the pure S^2 operator: + the compact term, no alpha pass follows)
  if (pass_mask & 8) {
    ListsArgs g;
    fill_pass_args(c, s, mode, spin, ss, shift, &g);
    g.in = d_c;
    g.in_stride = in_stride;
    g.out = d_sigma;
    g.out_stride = out_stride;
    g.t4 = (cross && !alpha_pass) ? s->t4.as<double>() : nullptr;
    g.t4_ld = mb;
    g.vec_index = vec_index;
    const int var = mode == 1 ? 3 : (spin ? 2 : 1);
    const size_t shmem = s->lds_b.bytes;
    const unsigned grid = 8u * (unsigned)g.cpx * (unsigned)g.nblk;
    static std::atomic<size_t> granted[4][64];
    const int dev = c->device & 63;
    const void* fn = var == 1   ? reinterpret_cast<const void*>(&k_sigma_lists<1>)
                     : var == 2 ? reinterpret_cast<const void*>(&k_sigma_lists<2>)
                                : reinterpret_cast<const void*>(&k_sigma_lists<3>);
    if (shmem > 64 * 1024 && shmem > granted[var][dev].load(std::memory_order_relaxed)) {
      SQD_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
      granted[var][dev].store(shmem, std::memory_order_relaxed);
    }
    switch (var) {
      case 1: hipLaunchKernelGGL((k_sigma_lists<1>), dim3(grid), dim3(NT), shmem, c->stream, g); break;
      case 2: hipLaunchKernelGGL((k_sigma_lists<2>), dim3(grid), dim3(NT), shmem, c->stream, g); break;
      default: hipLaunchKernelGGL((k_sigma_lists<3>), dim3(grid), dim3(NT), shmem, c->stream, g); break;
    }
    SQD_HIP_CHECK(hipGetLastError());
  }
  // launch 4: alpha lists by rows, added onto the list pass's result together with the compact term
  if (alpha_pass && (pass_mask & 4)) {
    AlphaRowsArgs a;
    std::memset(&a, 0, sizeof(a));
    a.in = d_c;
    a.out = d_sigma;
    a.in_stride = in_stride;
    a.out_stride = out_stride;
    a.na = na;
    a.nb = nb;
    a.pw = alpha_panel_width(na, nb);
    a.npanel = (int)((nb + a.pw - 1) / a.pw);
    a.chunk = 64;
    if (const char* env = std::getenv("SQD_ALPHA_CHUNK")) a.chunk = std::min(64, std::max(1, std::atoi(env)));
    a.s_ptr = c->sp[0].s_ptr.as<int64_t>();
    a.d_ptr = c->sp[0].d_ptr.as<int64_t>();
    a.s_rec = c->sp[0].s_rec.as<SRec>();
    a.s_val = c->sp[0].s_val.as<double>();
    a.d_src = c->sp[0].d_src.as<uint32_t>();
    a.d_val = c->sp[0].d_val.as<double>();
    a.jT = c->sp[1].jT.as<double>();
    a.accum = (pass_mask & 8) ? 1 : 0;
    a.t4 = cross ? s->t4.as<double>() : nullptr;
    a.t4_ld = mb;
    a.cidx_a = s->side[0].cidx.as<int32_t>();
    a.cidx_b = s->side[1].cidx.as<int32_t>();
    a.stop = c->sigma_stop;
    a.vec_index = vec_index;
    const bool wide = ((nb & 1) == 0) && ((reinterpret_cast<uintptr_t>(d_c) & 15) == 0) &&
                      ((reinterpret_cast<uintptr_t>(d_sigma) & 15) == 0) && ((reinterpret_cast<uintptr_t>(a.jT.p) & 15) == 0) &&
                      (!indexed || (((in_stride | out_stride) & 1) == 0));
    const int64_t tasks = na * a.npanel;
    const unsigned grid = (unsigned)((tasks + 3) / 4);
    if (wide) hipLaunchKernelGGL((k_alpha_rows<true>), dim3(grid), dim3(256), 0, c->stream, a);
    else hipLaunchKernelGGL((k_alpha_rows<false>), dim3(grid), dim3(256), 0, c->stream, a);
    SQD_HIP_CHECK(hipGetLastError());
  }
  if (c->ev_after_sigma_kernel) {
    SQD_HIP_CHECK(hipEventRecord(c->ev_after_sigma_kernel, c->stream));
    c->ev_after_sigma_kernel = nullptr;
  }
  return SQD_OK;
}

}  // namespace sqd

#ifdef SQD_PHASE_CLOCK
// out[4 * 8]: per kernel variant the column sums over the workgroups' rows {requests, row from LDS, barrier 1, next row into
// LDS, epilogue, barrier 2} in 10 ns units
extern "C" __attribute__((visibility("default"))) int sqd_probe_clk_lists(unsigned long long* out, int reset) {
  static std::vector<unsigned long long> h((size_t)4 * sqd::LCLK_ROWS * sqd::LCLK_COLS);
  if (out) {
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(sqd::sqd_clk_lists), h.size() * 8) != hipSuccess) return -1;
    for (int k = 0; k < 4 * sqd::LCLK_COLS; ++k) out[k] = 0;
    for (int v = 0; v < 4; ++v)
      for (size_t r = 0; r < (size_t)sqd::LCLK_ROWS; ++r)
        for (int k = 0; k < sqd::LCLK_COLS; ++k) out[v * sqd::LCLK_COLS + k] += h[(v * sqd::LCLK_ROWS + r) * sqd::LCLK_COLS + k];
  }
  if (reset) {
    std::fill(h.begin(), h.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(sqd::sqd_clk_lists), h.data(), h.size() * 8) != hipSuccess) return -1;
  }
  return 0;
}
#endif
