// Internal declarations shared by the HIP translation units of libsqd_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "sqd_hip.h"

namespace sqd {

// ---- error plumbing ---------------------------------------------------------
void set_error(const std::string& msg);
#define SQD_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      sqd::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + \
                     ":" + std::to_string(__LINE__));                                    \
      return SQD_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)
#define SQD_TRY(expr)          \
  do {                         \
    int _rc = (expr);          \
    if (_rc != SQD_OK) return _rc; \
  } while (0)

// ---- host waits.  hipStreamSynchronize / hipEventSynchronize go to sleep on an interrupt inside the runtime and wake
// up 10-20 us after the work is done.  (The 40-60 ms stalls round 1 saw are something else: the device settling its
// power state in a process' first ~0.1 s of activity, profiles/r02/stall_probe.txt.)  The solves here are
// sub-millisecond, so the host polls instead (hipStreamQuery / hipEventQuery never block) and only
// falls back to the blocking call after two seconds.
int spin_stream_sync(hipStream_t s);
// wait until a device-written sequence word in host-visible memory reaches `seq` (falls back to a stream sync)
int spin_wait_word(const void* word, long long seq, hipStream_t s);
int spin_event_sync(hipEvent_t e);
#define SQD_STREAM_SYNC(s) SQD_TRY(sqd::spin_stream_sync(s))

// ---- grow-only device buffer (arena semantics: reused across set_subspace calls) ----
struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  bool view = false;          // true: p points into another DevBuf (never freed or grown here)
  int reserve(size_t bytes);  // contents are NOT preserved on growth
  void release();
  void set_view(void* q) {
    if (!view) release();
    p = q;
    cap = 0;
    view = true;
  }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// ---- batched solves (sqd_solve_batch): one host-to-device copy per phase carries the kernel arguments of every
// subspace, the strings and the descriptor blobs: pinned host arena + its device twin, grow-only
struct BatchStage {
  char* host = nullptr;
  size_t cap = 0;
  DevBuf dev;
  int reserve(size_t bytes);  // contents are NOT preserved on growth
  void release();
};

// ---- pointers in kernel-argument records.  The batched kernels (sqd_solve_batch) read their arguments from device
// memory, and a plain pointer LOADED from memory is a generic ("flat") pointer to the compiler: every access through it
// becomes flat_load / flat_store, which also ties up the LDS counter (measured: the batched sigma kernel 1.5x slower per
// workgroup than the same body with by-value arguments).  A field declared GPtr<T> is typed as a global-address-space
// pointer in device code, so loads through it are global_load (s_load when uniform) wherever the record lives; in host
// code -- and to every reader -- it is a T*.  Same size and layout as T*.
template <class T>
struct GPtr {
#if defined(__HIP_DEVICE_COMPILE__) && __HIP_DEVICE_COMPILE__
  T __attribute__((address_space(1)))* p;
  __host__ __device__ GPtr& operator=(T* q) {
    p = (T __attribute__((address_space(1)))*)q;
    return *this;
  }
  __host__ __device__ operator T*() const { return (T*)p; }
#else
  T* p;
  __host__ __device__ GPtr& operator=(T* q) {
    p = q;
    return *this;
  }
  __host__ __device__ operator T*() const { return p; }
#endif
};

// dense same-spin mode: the matrix-core product is cut into this many partial products over disjoint k ranges (one
// workgroup each per 64 x 64 tile): a workgroup's chain of dependent loads is that much shorter and a single subspace of
// batch size (25 tiles at 317 x 317) still gives every CU work.  Fixed => the same bits in single and batched solves.
constexpr int DENSE_SPLIT = 8;

// ---- link record encodings ----------------------------------------------------
// single-excitation record: {src address, meta}
//   meta bits  0..12 : widx = 2*pair + dir   (pair = tril index of (cre,des); dir = cre > des)
//              13..18 : cre (created orbital), 19..24 : des (annihilated orbital), 31 : sign (1 = -1)
struct SRec {
  uint32_t src;
  uint32_t meta;
};
__host__ __device__ inline uint32_t srec_widx(uint32_t m) { return m & 0x1fffu; }
__host__ __device__ inline uint32_t srec_cre(uint32_t m) { return (m >> 13) & 63u; }
__host__ __device__ inline uint32_t srec_des(uint32_t m) { return (m >> 19) & 63u; }
__host__ __device__ inline double srec_sign(uint32_t m) { return (m >> 31) ? -1.0 : 1.0; }
// double-excitation orbital word: p | r<<6 | q<<12 | s<<18 | sign<<31
__host__ __device__ inline uint32_t tril(uint32_t p, uint32_t q) {
  return p >= q ? p * (p + 1) / 2 + q : q * (q + 1) / 2 + p;
}

// Per-spin tables.  "row" role = alpha (rows of C), "col" role = beta (columns of C); both
// spins carry the CSR form, the sliced-ELL copies are what the column role reads coalesced.
struct SpinTables {
  int64_t n = 0;
  int nocc = 0;
  int64_t n_s = 0, n_d = 0;      // populated single / double links
  int64_t n_slices = 0;          // ceil(n/64)
  DevBuf strs;                   // u64[n]
  DevBuf e_str;                  // f64[n]   same-spin diagonal energy of each string
  DevBuf s_ptr, d_ptr;           // i64[n+1] CSR row pointers
  DevBuf s_row, d_row;           // u32      COO row (tgt) of each link
  DevBuf s_rec;                  // SRec[n_s]
  DevBuf s_val;                  // f64[n_s] sign*(h_ab + sum_k in src (ab|kk)-(ak|kb))
  DevBuf d_src;                  // u32[n_d]
  DevBuf d_orb;                  // u32[n_d]
  DevBuf d_val;                  // f64[n_d] sign*((pq|rs)-(ps|rq))
  DevBuf jd_src, jd_val;         // the doubles in per-slice jagged-diagonal order (k_sigma_rows; beta only)
  DevBuf hs_ptr, hs_src, hs_val; // merged same-spin CSR {src, value}: singles then doubles per row (row role)
  DevBuf jrow;                   // f64[n][nnorb]   J[I][pair] = sum_{k in I} (pair|kk)   (row role)
  DevBuf jT;                     // f64[nnorb][n]   transposed copy                        (col role)
  // sliced ELL (slice = 64 consecutive strings), records of slice b start at *_sl[b], entry
  // (k, lane) lives at *_sl[b] + k*64 + lane
  DevBuf es_sl, ed_sl;           // i64[slices+1] over the virtual rows
  // capped ELL: every list is cut into virtual rows of <= cap links, ordered so that a wavefront sees
  // rows of equal length: all full rows first (grouped by owner string), then the tails by descending
  // length.  own[3B..3B+2] = {first full row, number of full rows, tail row or -1} of string B.
  int cap = 8;
  int64_t nv_s = 0, nv_d = 0;
  DevBuf vs_cnt, vs_own, vs_start, vd_cnt, vd_own, vd_start;  // i32[nv] / i32[3n] / i64[nv]
  // column chunks (one unless the row is too long for LDS): virtual rows of chunk k are
  // [v*_chunk[k], v*_chunk[k+1]), all owned by strings of that chunk
  DevBuf vs_chunk, vd_chunk;     // i32[nchunks+1]
  DevBuf es_rec;                 // SRec
  DevBuf es_val;                 // f64
  DevBuf ed_src;                 // u32
  DevBuf ed_val;                 // f64
  void release();
};

// One workgroup's worth of sigma work (see build_sigma_work in sqd_tables.hip)
struct WorkItem {
  int64_t begin;   // first link: index into the alpha singles arrays (type 1) or the merged hs arrays (0, 2)
  uint32_t A;      // alpha string = row of sigma
  uint16_t type;   // 0 own row, 1 alpha-single batch, 2 same-spin AXPY chunk
  uint16_t count;  // links in this item
  int32_t slot;    // -1: write sigma[A,:] directly; else partial row index
  int32_t pad;
};
struct MultiRow {
  uint32_t A;
  int32_t slot0, nslots;
};
// host copy of the capped-ELL descriptors (see sqd_tables.hip)
struct VRowsHost {
  std::vector<int32_t> vcnt, own, chunk;
  std::vector<int64_t> vstart, sl;
  int64_t nv = 0, total = 0, nv_max = 0;  // nv_max: most virtual rows in one column chunk
};

}  // namespace sqd

struct sqd_ctx {
  int device = 0;
  int norb = 0, nnorb = 0;
  int num_cu = 256;
  int lds_bytes = 160 * 1024;
  hipStream_t stream = nullptr;
  bool owns_stream = true;  // false after sqd_ctx_use_stream: the caller's stream is never destroyed here
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> sig_ev;  // (start, stop) pairs bracketing the sigma launches of a Davidson run
  // integrals
  sqd::DevBuf h1, eri4, eri_pp, jm, km;  // eri_pp[nnorb][nnorb]; jm/km[norb][norb]
  sqd::DevBuf jdiag;                     // [nnorb][norb]: (pq|kk)
  // subspace
  bool have_subspace = false;
  int64_t na = 0, nb = 0, D = 0;
  // SURVEY 8f-3, intra-solve sharding: this context produces sigma (and holds hdiag) only for the alpha rows
  // [row0, row1) of the subspace; the input vector is always the full na x nb matrix.  Default: all rows.
  int64_t row0 = 0, row1 = 0;
  bool sharded() const { return row0 != 0 || row1 != na; }
  int nelec[2] = {0, 0};
  sqd::SpinTables sp[2];
  sqd::DevBuf hdiag;        // f64[D]
  sqd::DevBuf strs2;        // u64[na + nb]: both string lists (SpinTables::strs are views)
  sqd::DevBuf guess_min;    // per alpha row: lowest diagonal element f64[na] and its flat index i64[na] (init guess)
  // sigma work list + launch geometry (fixed per subspace)
  sqd::DevBuf d_blob;               // packed capped-ELL descriptors (one upload); views in SpinTables
  // pinned host staging for the small uploads / downloads of set_subspace (a copy from pageable memory
  // blocks the host for ~10 us each; from pinned memory it is an asynchronous enqueue).  Bump-allocated,
  // reset at the start of every set_subspace once the previous one's copies have completed.
  std::vector<void*> stage_blocks;
  char* stage_cur = nullptr;
  size_t stage_cap = 0, stage_off = 0, stage_total = 0;
  bool stage_pending = false;   // uploads of the last set_subspace may still read the arena (cleared by any full sync)
  bool want_timing = false;     // record the set_subspace / Davidson phase events this call (stream bubbles: off unless asked)
  bool phase_timing = false;    // sticky request for the above (sqd_ctx_set_phase_timing)
  sqd::DevBuf ptrs;                 // [s_ptr_a | d_ptr_a | s_ptr_b | d_ptr_b]; SpinTables::s_ptr/d_ptr are views
  // host-visible (mapped, coherent) twin of `ptrs`: the scan writes every pointer to both, so the host reads them
  // without a copy command or an event; grow-only
  int64_t* h_ptrs_map = nullptr;
  int64_t* d_ptrs_map = nullptr;
  size_t ptrs_map_cap = 0;
  // (the host copy of the same lives in the pinned staging arena; h_sptr.. point into it)
  const int64_t *h_sptr = nullptr, *h_dptr = nullptr, *h_sptr_b = nullptr, *h_dptr_b = nullptr;
  std::vector<sqd::WorkItem> h_items;
  std::vector<sqd::MultiRow> h_multi;
  std::vector<int32_t> h_rowinfo;   // per alpha row {first partial slot, number of slots} (0 slots: sigma was written directly)
  sqd::DevBuf rowinfo;
  bool sigma_defer_reduce = false;  // Davidson: the fixed-order sum of a split row's partial rows is done by the
                                    // consumer of the new sigma vector (k_dots_eig) instead of a k_sigma_reduce launch
  sqd::VRowsHost hv_s, hv_d;
  sqd::DevBuf items, multi, sig_partial;
  int64_t n_items = 0, n_multi = 0, n_slots = 0;
  int sig_T = 64, sig_R = 1, sig_K = 1, sig_nb_pad = 0;
  int sig_kmax = 4;              // batch size limit chosen by the layout search in build_subspace
  size_t sig_shmem = 0;
  bool sig_lds_rows = true;      // C rows staged in LDS (false: rows too long, read from global/L2)
  // set_subspace's last launch leaves the state block + pyscf's start vector in X[0] (guess_x == X.p, one-shot)
  double* guess_x = nullptr;
  int dav_nvecs_hint = 13;       // vectors of the last Davidson workspace (12 + 1 by default)
  int sig_rows = 0;              // > 0: k_sigma_rows with this many rows of C per workgroup (implies sig_direct's
                                 // table layout: CSR lists only)
  bool sig_direct = false;       // ultra-sparse coupling: the element-gather kernel k_sigma_direct, no work items
  // large sets with short, even lists (10^4 x 10^4): k_sigma_lists -- link lists in registers, rows of C / C^T through
  // LDS, one pass per spin (sqd_lists.hip).  Implies sig_direct's table layout (CSR lists only) and sig_rows == 0.
  bool sig_lists = false;
  void* lists = nullptr;         // sqd::ListState (sqd_lists.hip), created on first use, released with the context
  // well-connected string sets (same-spin blocks >= ~8 % dense: what the SQD loop's carry-over produces, 20-22 %
  // measured, profiles/r03/loop_subspaces_probe.txt): the same-spin part H_a C + C H_b runs on the f64 matrix cores
  // (k_same_spin_mfma) from dense, zero-padded copies of the two symmetric blocks; the work items keep the
  // opposite-spin terms
  bool sig_dense = false;
  int dense_pa = 0, dense_pb = 0;        // padded orders (multiples of 64) = leading dimensions
  sqd::DevBuf hdense_a, hdense_b, gdense;  // f64[pa*pa], f64[pb*pb], f64[rows*nb] (the product, added by the own-row items)
  // connected sets of ~10^3 strings per spin and more (same-spin blocks 5-11 % dense): the same-spin part as a sparse
  // product in row-AXPY form on C and on C^T (sqd_spmm.hip) instead of the matrix cores; sig_dense is set as well (the
  // work items add ONE partial product, gdense) and hdense_a / hdense_b are not built
  bool sig_spmm = false;
  void* spmm = nullptr;          // sqd::SpmmState (sqd_spmm.hip), created on first use, released with the context
  // ... and the opposite-spin part + diagonal by whole rows with the beta link list in registers (sqd_opp.hip) instead of
  // the work items, for the plain operator (H without a spin penalty; S^2 and the penalty forms keep the work items)
  bool sig_opp = false;
  void* opp = nullptr;           // sqd::OppState (sqd_opp.hip)
  bool opp_src = false;          // ... by passes over source-column ranges (k_opp_src) instead of k_opp_rows
  void* oppsrc = nullptr;        // sqd::OppSrcState (sqd_oppsrc.hip)
  // row-sharded Davidson, sigma in two launches around the all-gather (shard_dav_sigma_part): 0 = one launch, 1 = the part
  // that needs only this rank's rows (input: sig_c_own), 2 = the rest; read by fill_sigma_args / launch_sigma
  int sig_part = 0;
  const double* sig_c_own = nullptr;
  int64_t sig_chunk = 0;         // columns per chunk (>= nb when there is one chunk)
  int sig_nchunks = 1;
  // LDS capacity (in virtual rows) of the singles' / doubles' partial-sum arrays; a chunk with more
  // virtual rows than that is walked in several passes over the same staged row
  int64_t sig_ps = 0, sig_pd = 0;
  // Davidson workspace
  sqd::DevBuf X, AX;        // (max_space+1) * D each
  sqd::DevBuf sol;          // f64[D] resident solution
  bool have_solution = false;
  // batched solves keep the PREVIOUS call's solution of each sub-context as well (sol and sol_prev swap at the start
  // of sqd_solve_batch): a caller that still holds the results of call N while call N + 1 runs -- `results =
  // solver(...)` in a loop -- can read their states afterwards without every one of them being copied out first
  sqd::DevBuf sol_prev;
  int64_t D_prev = 0;
  sqd::DevBuf tmp1, tmp2;   // f64[D] scratch vectors
  sqd::DevBuf io_in, io_out;  // staging of host vectors crossing the C ABI
  sqd::DevBuf partial;      // reduction partials
  sqd::DevBuf scal;         // small device scalars
  sqd::DevBuf scratch;      // misc (counts, scans)
  double* h_pinned = nullptr;  // pinned host scratch (>= 4096 doubles)
  // host-visible mailbox (fine-grained pinned memory): small reductions are written here by the device
  // and the host spins on the sequence word instead of paying a copy + stream synchronisation
  double* h_mail = nullptr;    // host pointer; [0] = sequence word (as int64), [8..] = payload
  double* d_mail = nullptr;    // the same memory as seen from the device
  int64_t mail_seq = 0;
  int64_t obs_seq = 0;  // sequence number the latest k_observables posts behind its results
  // sqd_ctx_set_async_state: sqd_solve returns when the RESULTS are on the host; the state follows (written by the
  // second stage of the observables kernel) and has landed when the mailbox's state word reaches state_seq
  bool async_state = false;
  int64_t state_seq = 0;
  // ... read by k_state_copy from `sol` while the NEXT solve already runs: that one forms its solution in sol_alt (the
  // two swap at the start of every asynchronous solve; sol_alt_ticket = the copy that may still be reading sol_alt)
  sqd::DevBuf sol_alt;
  int64_t sol_alt_ticket = 0, sol_ticket = 0;
  int64_t sigma_launches = 0;  // sigma launches of Davidson runs on this context (event sampling)
  double ms_setup = 0.0;
  std::vector<double> host_tmp;
  // sqd_solve: amplitudes travel to a pinned staging buffer on their own stream while the observables'
  // kernels run on the compute stream
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_sol = nullptr;
  hipEvent_t ev_after_sigma_kernel = nullptr;  // if set: recorded once, right after the next k_sigma launch
  const int* sigma_stop = nullptr;  // device flag honoured by the sigma launches of a Davidson run, else null
  const int* sigma_index = nullptr; // device word naming the basis vector a Davidson sigma works on (1-based), else null
  std::vector<int> dav_ev_iter;     // iteration index of each timed sigma launch of the latest run
  hipEvent_t ev_aux = nullptr;  // set_subspace: "CSR pointers are on the host" (later kernels keep running)
  double* h_amps = nullptr;
  size_t h_amps_cap = 0;
  bool dav_timed = false;  // the latest Davidson run recorded its start / end events
  int dav_nev = 0;  // timed sigma launches of the latest Davidson run (stats are collected after the sync)
  // ---- row-sharded Davidson in progress (sqd_shard_dav_*): constants of the run, the all-reduce buffers
  bool shard_active = false;
  bool shard_send_fresh = false;  // the send buffer holds the vector the next sigma build reads (left there by the orth stage)
  int shard_max_space = 12, shard_form = 0;
  double shard_prm_tol = 1e-9, shard_prm_tol2 = 0.0, shard_prm_lindep = 1e-14, shard_ss = 0.0, shard_shift = 0.0;
  int64_t shard_Dl = 0;
  sqd::DevBuf shard_tot;
  // ---- batched solves (sqd_solve_batch).  A parent context owns one sub-context per subspace of the batch: the
  // per-subspace state (tables, Davidson workspace, state block, mailbox) in the same struct a single solve uses, on
  // the PARENT's stream and with views of the parent's integral tables.
  // device address where the observables kernel of a solve also leaves its RAW record {e_davidson, c.Hc, c.S2c, c.c,
  // occ_a[norb], occ_b[norb], |S2 c|^2} (sqd_ctx_set_record_out): the input of a collective exchange that follows on
  // the same stream, no copy in between.  Batch p of sqd_solve_batch: record_out + p * record_stride.
  double* record_out = nullptr;
  // called by the solve calls between their last launch and their final wait (sqd_ctx_set_enqueue_hook)
  void (*enqueue_hook)(void*) = nullptr;
  void* enqueue_hook_user = nullptr;
  int64_t record_stride = 0;
  sqd_ctx* parent = nullptr;        // set on a sub-context
  std::vector<sqd_ctx*> subs;       // grow-only; subs[i] serves batch i of the latest sqd_solve_batch
  int batch_n = 0;                  // subspaces of the latest sqd_solve_batch (0: none)
  int batch_n_prev_valid = 0;       // ... of the latest one that completed (their solutions are resident)
  sqd::BatchStage bstage[3];        // [0] table build phase 1, [1] phase 2, [2] solver (sigma / Davidson / observables)
};

namespace sqd {
// tables (sqd_tables.hip)
int build_integral_tables(sqd_ctx* c, const double* h1, const double* eri);
int build_subspace(sqd_ctx* c, const uint64_t* sa, int64_t na, const uint64_t* sb, int64_t nb, int64_t row0 = 0,
                   int64_t row1 = -1);
int build_subspace_batch(sqd_ctx* parent, const std::vector<sqd_ctx*>& subs, const uint64_t* const* sa,
                         const int64_t* na, const uint64_t* const* sb, const int64_t* nb);
// sigma (sqd_sigma.hip).  mode 0: H (+ shift*(S^2-ss) if spin) ; mode 1: pure S^2
// in_stride / out_stride != 0 (inside a Davidson run): the vector is chosen on the device through
// sqd_ctx::sigma_index, input d_c + (*index - 1) * in_stride, output d_sigma + (*index - 1) * out_stride
int launch_sigma(sqd_ctx* c, const double* d_c, double* d_sigma, int mode, bool spin, double ss, double shift,
                 int64_t in_stride = 0, int64_t out_stride = 0);
int apply_h(sqd_ctx* c, const double* d_c, double* d_sigma, int use_spin, double ss, double shift, int64_t in_stride = 0,
            int64_t out_stride = 0);
int time_dense_product(sqd_ctx* c, const double* d_c, int reps, int copies, double* ms, double* flops);
// list-pass sigma for large sets with short lists (sqd_lists.hip).  lists_select: phase 2 of set_subspace, CSR pointers
// on the host, decides sqd_ctx::sig_lists and plans the column blocks; lists_build: device tables, behind launch C
bool lists_select(sqd_ctx* c, int64_t na, int64_t nb, int64_t row0, int64_t row1, const int64_t* tot, const int* nocc);
int lists_build(sqd_ctx* c);
int launch_sigma_lists(sqd_ctx* c, const double* d_c, double* d_sigma, int mode, bool spin, double ss, double shift,
                       int64_t in_stride, int64_t out_stride);
void lists_release(sqd_ctx* c);
// sparse-product same-spin part for connected sets of ~10^3 strings per spin and more (sqd_spmm.hip).  spmm_select:
// phase 2 of set_subspace (sets sqd_ctx::sig_spmm); spmm_build: device tables, behind launch C; spmm_launch: G =
// H_a C + C H_b into sqd_ctx::gdense, in front of the work items of the same sigma build
bool spmm_select(sqd_ctx* c, int64_t na, int64_t nb, int64_t row0, int64_t row1, const int64_t* tot, bool direct);
int spmm_build(sqd_ctx* c);
int spmm_launch(sqd_ctx* c, const double* d_c, int64_t in_stride);
void spmm_release(sqd_ctx* c);
int spmm_rows_per_group(const sqd_ctx* c);  // 8: the row-grouped product (k_spmm_grouped); 1: one row per wavefront (k_spmm_rows)
// opposite-spin part + diagonal by whole rows for the subspaces of the sparse-product path (sqd_opp.hip).  opp_select:
// phase 2 of set_subspace, behind spmm_select (sets sqd_ctx::sig_opp); opp_build: device tables, behind launch C;
// opp_launch: sigma = (hdiag + opposite-spin part) c + gdense, behind spmm_launch of the same vector
bool opp_select(sqd_ctx* c, int64_t na, int64_t nb, const int64_t* tot);
int opp_build(sqd_ctx* c);
int opp_launch(sqd_ctx* c, const double* d_c, double* d_sigma, int64_t in_stride, int64_t out_stride, bool spin = false,
               double ss = 0.0, double shift = 0.0);
void opp_release(sqd_ctx* c);
bool opp_split(const sqd_ctx* c, const int32_t** rowinfo, const double** partial);
// ... for rows of more than 3072 columns: passes over ranges of the SOURCE column (sqd_oppsrc.hip); reached through the
// opp_* entry points above (sqd_ctx::opp_src)
bool oppsrc_select(sqd_ctx* c, int64_t na, int64_t nb, const int64_t* tot);
int oppsrc_build(sqd_ctx* c);
int oppsrc_launch(sqd_ctx* c, const double* d_c, double* d_sigma, int64_t in_stride, int64_t out_stride, bool spin, double ss, double shift);
void oppsrc_release(sqd_ctx* c);
bool oppsrc_split(const sqd_ctx* c, const int32_t** rowinfo, const double** partial);
int oppsrc_passes(const sqd_ctx* c);
// batched sigma (sqd_solve_batch): per launch class one launch over all subspaces of the class
struct SigmaBatchPlan {
  struct Launch {
    int kind;  // 0 work items, 1 element gather, 3 the fixed-order sum of split rows
    int R;
    bool spin;
    unsigned gx, gy;
    int T, n;
    size_t shmem;
    const char* args;  // device array of the class' argument structs
  };
  std::vector<Launch> launches;
};
bool sigma_batch_supported(const sqd_ctx* c);
size_t sigma_batch_bytes(size_t nsub);
// stride_scale != 0: inside a Davidson run (vector chosen on the device, strides = each subspace's D)
int sigma_batch_plan(const std::vector<sqd_ctx*>& subs, const std::vector<const double*>& d_c,
                     const std::vector<double*>& d_sigma, int mode, bool spin, double ss, double shift,
                     int64_t stride_scale, char* h, char* d, size_t* off_io, SigmaBatchPlan* plan,
                     bool skip_direct = false);  // skip_direct: the caller launches the element-gather class itself
int sigma_batch_launch(sqd_ctx* parent, const SigmaBatchPlan& plan);
// blas-1 (sqd_davidson.hip)
int dev_dot(sqd_ctx* c, const double* x, const double* y, double* out);
int enqueue_init_guess(sqd_ctx* c, double* d_x);  // pyscf get_init_guess into d_x (no synchronisation)
// defer_sync: return with the solution still being formed on the stream; the caller synchronises and
// then calls davidson_collect, which fills *st (outcome + event timings)
int run_davidson(sqd_ctx* c, const sqd_davidson_opts* o, const double* ci0_host, sqd_davidson_stats* st,
                 bool defer_sync = false);
int davidson_collect(sqd_ctx* c, sqd_davidson_stats* st);
// row-sharded Davidson, stage by stage (sqd_shard_dav_* of the C ABI)
int shard_dav_begin(sqd_ctx* c, const sqd_davidson_opts* o, double** d_x0);
int shard_dav_pick(sqd_ctx* c, double** d_send);
int shard_dav_sigma(sqd_ctx* c, const double* d_full, int part = 0);
int shard_dav_dots(sqd_ctx* c, double** d_tot, int* count);
int shard_dav_residual(sqd_ctx* c, double** d_tot2, int* count, bool eig_done = false);
int shard_dav_iteration(sqd_ctx* c, long long* seq_out);
int shard_dav_orth(sqd_ctx* c, long long* seq_out);
int shard_dav_wait(sqd_ctx* c, long long seq, int* stopped, double* e, double* rnorm2, int* m_cur);
int shard_dav_end(sqd_ctx* c, double** d_solution_rows, sqd_davidson_stats* st);
// batched Davidson (sqd_solve_batch): prepare writes the per-subspace argument records (and the sigma plan) into the
// staging blob at *off_io; run enqueues rounds until every subspace has stopped, then the solutions
struct DavBatchPlan {
  const char* args = nullptr;  // device array of DavBatchArgs
  int n = 0, max_space = 12, max_cycle = 100;
  unsigned gb = 1;
  SigmaBatchPlan sigma;
  // subspaces of the element-gather class: sigma build and dot products in ONE launch (k_sigma_dots_b), as the single solve
  const char* fused_args = nullptr;  // device array of their DirectArgs
  const char* fused_map = nullptr;   // ... and of their indices into `args`
  int n_fused = 0;
  bool fused_spin = false;
};
size_t davidson_batch_bytes(size_t nsub);
int davidson_batch_prepare(sqd_ctx* parent, const std::vector<sqd_ctx*>& subs, const sqd_davidson_opts* o, char* h,
                           char* d, size_t* off_io, DavBatchPlan* plan);
int davidson_batch_run(sqd_ctx* parent, const std::vector<sqd_ctx*>& subs, const DavBatchPlan& plan);
// arrival counters shared by the fused ("last workgroup finishes") reductions of a context's stream.  They reset
// themselves after every use; a Davidson run also zeroes them, so a kernel aborted mid-way cannot poison later ones.
int reserve_counters(sqd_ctx* c);
unsigned* counter_ptr(sqd_ctx* c);
unsigned* counter2_ptr(sqd_ctx* c);  // a second, independent set (same self-resetting protocol): k_state_copy
unsigned* counter3_ptr(sqd_ctx* c);  // a third: k_tables_diag_fill
void* dav_state_ptr(sqd_ctx* c);  // DavState* (sqd_davstate.h)
// observables (sqd_rdm.hip)
int dev_rdm1s(sqd_ctx* c, const double* d_c, double* dm1a, double* dm1b);
int dev_rdm2(sqd_ctx* c, const double* d_c, double* dm2);
int dev_rdm2s(sqd_ctx* c, const double* d_c, double* dm2aa, double* dm2ab, double* dm2bb);
int dev_observables(sqd_ctx* c, const double* d_c, double* out_host);
// kernels only, no synchronisation.  with_h: <c|H|c> by a sigma build (else out[0] = 0); with_s2: S^2 c is built and
// <c|S^2|c>, |S^2 c|^2 reduced (else 0).  Results land in host-visible memory, read by dev_observables_collect:
// out = {c.Hc, c.S2c, c.c, occ_a[norb], occ_b[norb], |S2 c|^2}
// late_state: the copy of the state into host_twin is a SECOND stage of the kernel, behind the results and their sequence
// word (sqd_ctx::state_seq is posted when it has landed): dev_observables_wait(c, false) returns with the results alone
int dev_observables_enqueue(sqd_ctx* c, const double* d_c, bool with_h = true, bool with_s2 = true,
                            double* host_twin = nullptr, bool late_state = false);
int dev_observables_wait(sqd_ctx* c, bool whole_kernel = true);
int state_copy_wait(sqd_ctx* c, long long ticket);
int state_copy_enqueue(sqd_ctx* c, double* host_twin, long long* ticket);  // k_state_copy on the copy stream
bool state_copy_landed(const sqd_ctx* c, long long ticket);
int sol_writer_guard(sqd_ctx* c);  // wait until no k_state_copy reads c->sol any more (every writer of sol calls it)
// batched (sqd_solve_batch)
struct ObsBatchPlan {
  const char* args = nullptr;  // device array of ObsArgs
  int n = 0;
  unsigned gx = 1;
  bool have_s2_sigma = false;
  SigmaBatchPlan s2_sigma;
};
size_t observables_batch_bytes(size_t nsub);
int observables_batch_prepare(sqd_ctx* parent, const std::vector<sqd_ctx*>& subs, bool with_s2,
                              const std::vector<double*>& host_twin, char* h, char* d, size_t* off_io,
                              ObsBatchPlan* plan);
int observables_batch_launch(sqd_ctx* parent, const ObsBatchPlan& plan);
void dev_observables_collect(sqd_ctx* c, double* out_host);   // after the stream has been synchronised
}  // namespace sqd
