// extern "C" boundary of libsqd_hip.so (declared in include/sqd_hip.h).
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>

#include <algorithm>

#include "sqd_common.h"

namespace sqd {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace sqd

namespace sqd {
// Host waits.  Everything on the solve path of a batch-sized subspace is sub-millisecond, and a blocking runtime wait
// wakes up 10-20 us late, so a wait first SPINS -- on the memory word the awaited kernel writes, or on the stream's
// status -- but only for SPIN_US: a solve of 1e7-1e8 determinants waits for milliseconds to seconds per round, and a
// thread that spins through that burns a core and starves the launches of the other solver threads.  Past the budget
// the wait sleeps in short naps (word waits) or hands over to the runtime's blocking call.
static inline double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
constexpr double SPIN_US = 300.0;
int spin_stream_sync(hipStream_t s) {
  const double t0 = now_us();
  for (long spin = 0;; ++spin) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return SQD_OK;
    if (e != hipErrorNotReady) {
      set_error(std::string("hipStreamQuery: ") + hipGetErrorString(e));
      return SQD_ERR_HIP;
    }
    if ((spin & 15) == 15 && now_us() - t0 > SPIN_US) break;
    // (hipStreamQuery takes a runtime lock that other host threads need for their launches: ~1 us between polls)
    for (int p = 0; p < 64; ++p) __builtin_ia32_pause();
  }
  SQD_HIP_CHECK(hipStreamSynchronize(s));
  return SQD_OK;
}
int spin_wait_word(const void* word, long long seq, hipStream_t s) {
  volatile const long long* flag = reinterpret_cast<volatile const long long*>(word);
  const double t0 = now_us();
  for (long spin = 0;; ++spin) {
    if (*flag >= seq) {  // sequence numbers only grow on a context: a later post implies this one
      std::atomic_thread_fence(std::memory_order_acquire);
      return SQD_OK;
    }
    __builtin_ia32_pause();
    if ((spin & 255) == 255 && now_us() - t0 > SPIN_US) break;
  }
  // long wait: nap between looks at the word; the stream's status tells a kernel that died from one still running
  for (long nap = 0;; ++nap) {
    if (*flag >= seq) {
      std::atomic_thread_fence(std::memory_order_acquire);
      return SQD_OK;
    }
    if ((nap & 7) == 7) {
      const hipError_t e = hipStreamQuery(s);
      if (e == hipSuccess) break;  // everything enqueued has run: the word is there now, or never will be
      if (e != hipErrorNotReady) {
        set_error(std::string("hipStreamQuery: ") + hipGetErrorString(e));
        return SQD_ERR_HIP;
      }
    }
    std::this_thread::sleep_for(std::chrono::microseconds(20));
    if (now_us() - t0 > 120e6) break;  // two minutes: let the runtime report what is wrong
  }
  SQD_TRY(spin_stream_sync(s));  // also surfaces asynchronous kernel errors
  if (*flag < seq) {
    set_error("device mailbox was not written");
    return SQD_ERR_HIP;
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return SQD_OK;
}
int spin_event_sync(hipEvent_t ev) {
  const double t0 = now_us();
  for (long spin = 0;; ++spin) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return SQD_OK;
    if (e != hipErrorNotReady) {
      set_error(std::string("hipEventQuery: ") + hipGetErrorString(e));
      return SQD_ERR_HIP;
    }
    if ((spin & 15) == 15 && now_us() - t0 > SPIN_US) break;
    __builtin_ia32_pause();
  }
  SQD_HIP_CHECK(hipEventSynchronize(ev));
  return SQD_OK;
}
}  // namespace sqd

using namespace sqd;

#define SQD_API extern "C" __attribute__((visibility("default")))

SQD_API int sqd_abi_version(void) { return 3; }

// ---- page-locked host buffers for results.  A caller that hands sqd_solve an amplitude buffer obtained here gets the
// state written straight into it by the GPU (by the observables kernel up to 64 MB, by the DMA engine beyond): no
// staging copy, and no first-touch page faults of a fresh 0.8 MB numpy allocation per solve (together ~60 us of a
// 0.3 ms solve).
namespace {
std::mutex g_pin_mu;
struct PinnedBlock {
  size_t bytes;
  char* dev;  // the block's address as the devices see it (looked up once, at allocation)
};
std::map<const char*, PinnedBlock> g_pinned;  // start -> block
// device-side address of a host range if it lies inside one block of sqd_host_alloc, else nullptr
void* pinned_device_ptr(const void* p, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_pin_mu);
  auto it = g_pinned.upper_bound(static_cast<const char*>(p));
  if (it == g_pinned.begin()) return nullptr;
  --it;
  if (static_cast<const char*>(p) + bytes > it->first + it->second.bytes) return nullptr;
  return it->second.dev + (static_cast<const char*>(p) - it->first);
}
}  // namespace
SQD_API int sqd_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return SQD_ERR_INVALID;
  void* p = nullptr;
  // (portable + mapped: every device of the process may write into it from a kernel, see sqd_solve)
  SQD_HIP_CHECK(hipHostMalloc(&p, bytes, hipHostMallocPortable | hipHostMallocMapped));
  void* dptr = nullptr;
  SQD_HIP_CHECK(hipHostGetDevicePointer(&dptr, p, 0));
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pinned[static_cast<const char*>(p)] = PinnedBlock{bytes, static_cast<char*>(dptr)};
  }
  *out = p;
  return SQD_OK;
}
SQD_API int sqd_host_free(void* p) {
  if (!p) return SQD_OK;
  {
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pinned.erase(static_cast<const char*>(p));
  }
  SQD_HIP_CHECK(hipHostFree(p));
  return SQD_OK;
}
SQD_API const char* sqd_last_error(void) { return g_err.c_str(); }

SQD_API int sqd_device_count(int* count) {
  if (!count) return SQD_ERR_INVALID;
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    set_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return SQD_ERR_HIP;
  }
  return SQD_OK;
}

SQD_API int sqd_ctx_create(int device, int norb, const double* h1, const double* eri, sqd_ctx** out) {
  if (!out || !h1 || !eri) {
    set_error("null argument");
    return SQD_ERR_INVALID;
  }
  if (norb < 1 || norb > SQD_MAX_NORB) {
    set_error("norb must be in [1, 64]");
    return SQD_ERR_INVALID;
  }
  SQD_HIP_CHECK(hipSetDevice(device));
  sqd_ctx* c = new sqd_ctx();
  c->device = device;
  c->norb = norb;
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && v > 0) c->num_cu = v;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && v > 0)
    c->lds_bytes = v;
  hipError_t e = hipStreamCreate(&c->stream);
  if (e != hipSuccess) {
    set_error(std::string("hipStreamCreate: ") + hipGetErrorString(e));
    delete c;
    return SQD_ERR_HIP;
  }
  e = hipStreamCreate(&c->copy_stream);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_sol);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_aux);
  if (e != hipSuccess) {
    set_error(std::string("hipStreamCreate/hipEventCreate: ") + hipGetErrorString(e));
    delete c;
    return SQD_ERR_HIP;
  }
  for (int i = 0; i < 4; ++i) {
    e = hipEventCreate(&c->ev[i]);
    if (e != hipSuccess) {
      set_error(std::string("hipEventCreate: ") + hipGetErrorString(e));
      delete c;
      return SQD_ERR_HIP;
    }
  }
  c->sig_ev.resize(4 * 128, nullptr);  // (start, after k_sigma, end, end of an empty bracket) per timed sigma application
  for (auto& evt : c->sig_ev) {
    e = hipEventCreate(&evt);
    if (e != hipSuccess) {
      set_error(std::string("hipEventCreate: ") + hipGetErrorString(e));
      delete c;
      return SQD_ERR_HIP;
    }
  }
  e = hipHostMalloc((void**)&c->h_pinned, 4096 * sizeof(double), hipHostMallocDefault);
  if (e != hipSuccess) {
    set_error(std::string("hipHostMalloc: ") + hipGetErrorString(e));
    delete c;
    return SQD_ERR_HIP;
  }
  e = hipHostMalloc((void**)&c->h_mail, 1024 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&c->d_mail, c->h_mail, 0);
  if (e != hipSuccess) {
    set_error(std::string("hipHostMalloc(mapped): ") + hipGetErrorString(e));
    delete c;
    return SQD_ERR_HIP;
  }
  std::memset(c->h_mail, 0, 1024 * sizeof(double));
  int rc = build_integral_tables(c, h1, eri);
  if (rc != SQD_OK) {
    sqd_ctx_destroy(c);
    return rc;
  }
  *out = c;
  return SQD_OK;
}

SQD_API int sqd_ctx_use_stream(sqd_ctx* c, void* stream) {
  if (!c) {
    set_error("sqd_ctx_use_stream: null context");
    return SQD_ERR_INVALID;
  }
  SQD_HIP_CHECK(hipSetDevice(c->device));
  if (c->stream) SQD_STREAM_SYNC(c->stream);  // nothing of ours is left in flight
  if (c->stream && c->owns_stream) SQD_HIP_CHECK(hipStreamDestroy(c->stream));
  c->stream = reinterpret_cast<hipStream_t>(stream);
  c->owns_stream = false;
  for (sqd_ctx* sub : c->subs) sub->stream = c->stream;  // sub-contexts of batched solves share the stream
  return SQD_OK;
}

SQD_API int sqd_ctx_destroy(sqd_ctx* c) {
  if (!c) return SQD_OK;
  hipError_t e = hipSetDevice(c->device);
  (void)e;
  if (c->stream) e = hipStreamSynchronize(c->stream);
  if (c->copy_stream) e = hipStreamSynchronize(c->copy_stream);  // (a state may still be leaving: k_state_copy)
  for (sqd_ctx* sub : c->subs) sqd_ctx_destroy(sub);  // (their integral tables are views of this context's)
  c->subs.clear();
  for (auto& bs : c->bstage) bs.release();
  DevBuf* bufs[] = {&c->h1, &c->eri4, &c->eri_pp, &c->jm, &c->km, &c->hdiag, &c->X, &c->AX,
                    &c->sol, &c->tmp1, &c->tmp2, &c->partial, &c->scal, &c->scratch, &c->io_in, &c->io_out,
                    &c->items, &c->multi, &c->sig_partial, &c->ptrs, &c->d_blob, &c->strs2, &c->guess_min, &c->jdiag, &c->rowinfo,
                    &c->hdense_a, &c->hdense_b, &c->gdense, &c->sol_prev, &c->shard_tot, &c->sol_alt};
  for (DevBuf* b : bufs) b->release();
  lists_release(c);
  spmm_release(c);
  opp_release(c);
  c->sp[0].release();
  c->sp[1].release();
  for (int i = 0; i < 4; ++i)
    if (c->ev[i]) e = hipEventDestroy(c->ev[i]);
  for (auto& evt : c->sig_ev)
    if (evt) e = hipEventDestroy(evt);
  if (c->h_pinned) e = hipHostFree(c->h_pinned);
  if (c->h_mail) e = hipHostFree(c->h_mail);
  if (c->h_amps) e = hipHostFree(c->h_amps);
  if (c->h_ptrs_map) e = hipHostFree(c->h_ptrs_map);
  for (void* p : c->stage_blocks) e = hipHostFree(p);
  if (c->ev_sol) e = hipEventDestroy(c->ev_sol);
  if (c->ev_aux) e = hipEventDestroy(c->ev_aux);
  if (c->copy_stream) e = hipStreamDestroy(c->copy_stream);
  if (c->stream && c->owns_stream) e = hipStreamDestroy(c->stream);
  delete c;
  return SQD_OK;
}

#define CTX_ENTER(c)                           \
  if (!(c)) {                                  \
    set_error("null context");                 \
    return SQD_ERR_INVALID;                    \
  }                                            \
  SQD_HIP_CHECK(hipSetDevice((c)->device));

#define NEED_SUBSPACE(c)              \
  if (!(c)->have_subspace) {          \
    set_error("no subspace set");     \
    return SQD_ERR_STATE;             \
  }
// entry points that work on whole vectors held by this context: not available on a row shard
#define NEED_ALL_ROWS(c)                                                                         \
  if ((c)->sharded()) {                                                                          \
    set_error("this context holds a row shard (sqd_set_subspace_rows): use the *_rows entry points"); \
    return SQD_ERR_STATE;                                                                        \
  }
// table inspection uses blocking copies: drain the context stream first (set_subspace returns
// without synchronising)
#define DRAIN(c)                  \
  SQD_STREAM_SYNC((c)->stream); \
  (c)->stage_pending = false

SQD_API int sqd_set_subspace(sqd_ctx* c, const uint64_t* strs_a, int64_t na, const uint64_t* strs_b, int64_t nb) {
  CTX_ENTER(c);
  c->want_timing = c->phase_timing;
  return build_subspace(c, strs_a, na, strs_b, nb);
}

SQD_API int sqd_set_subspace_rows(sqd_ctx* c, const uint64_t* strs_a, int64_t na, const uint64_t* strs_b, int64_t nb,
                                  int64_t row0, int64_t row1) {
  CTX_ENTER(c);
  return build_subspace(c, strs_a, na, strs_b, nb, row0, row1);
}

// sigma rows [row0, row1) from the FULL vector, both in device memory (device pointers, e.g. torch tensors'
// data_ptr()); enqueued on the context's stream, no synchronisation
SQD_API int sqd_sigma_rows_dev(sqd_ctx* c, const double* d_c_full, double* d_sigma_rows, int use_spin, double ss,
                               double shift) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  if (!d_c_full || !d_sigma_rows) return SQD_ERR_INVALID;
  if (use_spin == 2 || use_spin == 3) {
    // (S^2 - ss)^2 chains S^2 through intermediate FULL vectors: each application needs its own all-gather, which is
    // the caller's side of the exchange -- apply use_spin = 1 / sqd_contract_ss_rows_dev step by step instead
    const double sz = 0.5 * std::abs(c->nelec[0] - c->nelec[1]);
    if (use_spin == 2 || !(ss < sz * (sz + 1.0) + 0.1)) {
      set_error("the squared spin penalty is not available on a row shard in one call");
      return SQD_ERR_INVALID;
    }
    use_spin = 1;
  }
  return apply_h(c, d_c_full, d_sigma_rows, use_spin, ss, shift);
}
SQD_API int sqd_contract_ss_rows_dev(sqd_ctx* c, const double* d_c_full, double* d_out_rows) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  if (!d_c_full || !d_out_rows) return SQD_ERR_INVALID;
  return launch_sigma(c, d_c_full, d_out_rows, 1, false, 0.0, 0.0);
}
SQD_API int sqd_hdiag_rows_dev(sqd_ctx* c, double* d_out_rows) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  if (!d_out_rows) return SQD_ERR_INVALID;
  SQD_HIP_CHECK(hipMemcpyAsync(d_out_rows, c->hdiag.p, (size_t)(c->row1 - c->row0) * c->nb * 8, hipMemcpyDeviceToDevice,
                               c->stream));
  return SQD_OK;
}
SQD_API int sqd_solution_device_ptr(sqd_ctx* c, const double** d_ptr) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  if (!d_ptr) return SQD_ERR_INVALID;
  if (!c->have_solution) {
    set_error("no resident solution: run sqd_davidson / sqd_solve first");
    return SQD_ERR_STATE;
  }
  *d_ptr = c->sol.as<double>();
  return SQD_OK;
}
// ---- row-sharded Davidson (include/sqd_hip.h)
SQD_API int sqd_shard_dav_begin(sqd_ctx* c, const sqd_davidson_opts* opts, double** d_x0_rows) {
  CTX_ENTER(c);
  if (!d_x0_rows) return SQD_ERR_INVALID;
  sqd_davidson_opts o;
  if (opts) o = *opts; else sqd_davidson_default_opts(&o);
  if (o.tol <= 0 || o.max_cycle < 1) {
    set_error("bad Davidson options");
    return SQD_ERR_INVALID;
  }
  return shard_dav_begin(c, &o, d_x0_rows);
}
SQD_API int sqd_shard_dav_pick(sqd_ctx* c, double** d_send_rows) {
  CTX_ENTER(c);
  if (!d_send_rows) return SQD_ERR_INVALID;
  return shard_dav_pick(c, d_send_rows);
}
SQD_API int sqd_shard_dav_sigma(sqd_ctx* c, const double* d_c_full) {
  CTX_ENTER(c);
  if (!d_c_full) return SQD_ERR_INVALID;
  return shard_dav_sigma(c, d_c_full);
}
SQD_API int sqd_shard_dav_sigma_part(sqd_ctx* c, const double* d_c_full, int part) {
  CTX_ENTER(c);
  if (part < 0 || part > 2) return SQD_ERR_INVALID;
  if (part != 1 && !d_c_full) return SQD_ERR_INVALID;  // only part 1 (own rows, in front of the gather) reads no full vector
  return shard_dav_sigma(c, d_c_full, part);
}
SQD_API int sqd_shard_dav_dots(sqd_ctx* c, double** d_totals, int* count) {
  CTX_ENTER(c);
  if (!d_totals || !count) return SQD_ERR_INVALID;
  return shard_dav_dots(c, d_totals, count);
}
SQD_API int sqd_shard_dav_residual(sqd_ctx* c, double** d_totals, int* count) {
  CTX_ENTER(c);
  if (!d_totals || !count) return SQD_ERR_INVALID;
  return shard_dav_residual(c, d_totals, count);
}
SQD_API int sqd_shard_dav_iteration(sqd_ctx* c, long long* ticket) {
  CTX_ENTER(c);
  if (!ticket) return SQD_ERR_INVALID;
  return shard_dav_iteration(c, ticket);
}
SQD_API int sqd_shard_dav_orth(sqd_ctx* c, long long* ticket) {
  CTX_ENTER(c);
  return shard_dav_orth(c, ticket);
}
SQD_API int sqd_shard_dav_wait(sqd_ctx* c, long long ticket, int* stopped, double* e, double* rnorm2, int* basis_size) {
  CTX_ENTER(c);
  return shard_dav_wait(c, ticket, stopped, e, rnorm2, basis_size);
}
SQD_API int sqd_shard_dav_end(sqd_ctx* c, double** d_solution_rows, sqd_davidson_stats* stats) {
  CTX_ENTER(c);
  return shard_dav_end(c, d_solution_rows, stats);
}
SQD_API int sqd_solution_copy(sqd_ctx* c, double* amps) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  if (!amps) return SQD_ERR_INVALID;
  if (!c->have_solution) {
    set_error("no resident solution: run sqd_davidson / sqd_solve first");
    return SQD_ERR_STATE;
  }
  SQD_HIP_CHECK(hipMemcpyAsync(amps, c->sol.p, (size_t)c->D * 8, hipMemcpyDeviceToHost, c->stream));
  SQD_TRY(spin_stream_sync(c->stream));
  return SQD_OK;
}
SQD_API int sqd_ctx_set_record_out(sqd_ctx* c, double* d_record, int64_t stride) {
  if (!c) return SQD_ERR_INVALID;
  c->record_out = d_record;
  c->record_stride = stride;
  return SQD_OK;
}
SQD_API int sqd_ctx_set_enqueue_hook(sqd_ctx* c, sqd_enqueue_hook hook, void* user) {
  if (!c) return SQD_ERR_INVALID;
  c->enqueue_hook = hook;
  c->enqueue_hook_user = user;
  return SQD_OK;
}
SQD_API int sqd_ctx_set_async_state(sqd_ctx* c, int on) {
  CTX_ENTER(c);
  c->async_state = on != 0;
  return SQD_OK;
}
SQD_API int sqd_ctx_state_wait(sqd_ctx* c, long long ticket) {
  CTX_ENTER(c);
  if (ticket <= 0) return SQD_OK;
  return state_copy_wait(c, ticket);
}

SQD_API int sqd_ctx_set_phase_timing(sqd_ctx* c, int on) {
  if (!c) return SQD_ERR_INVALID;
  c->phase_timing = on != 0;
  return SQD_OK;
}
SQD_API int sqd_ctx_sync(sqd_ctx* c) {
  CTX_ENTER(c);
  SQD_STREAM_SYNC(c->stream);
  c->stage_pending = false;
  return SQD_OK;
}

SQD_API int sqd_get_dims(sqd_ctx* c, int64_t* na, int64_t* nb, int* nelec_a, int* nelec_b) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  if (na) *na = c->na;
  if (nb) *nb = c->nb;
  if (nelec_a) *nelec_a = c->nelec[0];
  if (nelec_b) *nelec_b = c->nelec[1];
  return SQD_OK;
}

SQD_API int sqd_link_counts(sqd_ctx* c, int spin, int64_t* n_single, int64_t* n_double) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  if (spin < 0 || spin > 1) return SQD_ERR_INVALID;
  if (n_single) *n_single = c->sp[spin].n_s;
  if (n_double) *n_double = c->sp[spin].n_d;
  return SQD_OK;
}

SQD_API int sqd_single_links(sqd_ctx* c, int spin, int32_t* tgt, int32_t* src, int32_t* cre, int32_t* des,
                             int32_t* pair, int32_t* sign, double* value) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  if (spin < 0 || spin > 1) return SQD_ERR_INVALID;
  const SpinTables& t = c->sp[spin];
  const int64_t n = t.n_s;
  if (n == 0) return SQD_OK;
  std::vector<SRec> rec(n);
  std::vector<uint32_t> row(n);
  DRAIN(c);
  SQD_HIP_CHECK(hipMemcpy(rec.data(), t.s_rec.p, n * sizeof(SRec), hipMemcpyDeviceToHost));
  SQD_HIP_CHECK(hipMemcpy(row.data(), t.s_row.p, n * 4, hipMemcpyDeviceToHost));
  if (value) SQD_HIP_CHECK(hipMemcpy(value, t.s_val.p, n * 8, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t m = rec[i].meta;
    if (tgt) tgt[i] = (int32_t)row[i];
    if (src) src[i] = (int32_t)rec[i].src;
    if (cre) cre[i] = (int32_t)srec_cre(m);
    if (des) des[i] = (int32_t)srec_des(m);
    if (pair) pair[i] = (int32_t)(srec_widx(m) >> 1);
    if (sign) sign[i] = (m >> 31) ? -1 : 1;
  }
  return SQD_OK;
}

SQD_API int sqd_double_links(sqd_ctx* c, int spin, int32_t* tgt, int32_t* src, int32_t* orbs, int32_t* sign,
                             double* value) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  if (spin < 0 || spin > 1) return SQD_ERR_INVALID;
  const SpinTables& t = c->sp[spin];
  const int64_t n = t.n_d;
  if (n == 0) return SQD_OK;
  std::vector<uint32_t> row(n), sr(n), ob(n);
  DRAIN(c);
  SQD_HIP_CHECK(hipMemcpy(row.data(), t.d_row.p, n * 4, hipMemcpyDeviceToHost));
  SQD_HIP_CHECK(hipMemcpy(sr.data(), t.d_src.p, n * 4, hipMemcpyDeviceToHost));
  SQD_HIP_CHECK(hipMemcpy(ob.data(), t.d_orb.p, n * 4, hipMemcpyDeviceToHost));
  if (value) SQD_HIP_CHECK(hipMemcpy(value, t.d_val.p, n * 8, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < n; ++i) {
    if (tgt) tgt[i] = (int32_t)row[i];
    if (src) src[i] = (int32_t)sr[i];
    if (orbs) {
      orbs[4 * i + 0] = ob[i] & 63;
      orbs[4 * i + 1] = (ob[i] >> 6) & 63;
      orbs[4 * i + 2] = (ob[i] >> 12) & 63;
      orbs[4 * i + 3] = (ob[i] >> 18) & 63;
    }
    if (sign) sign[i] = (ob[i] >> 31) ? -1 : 1;
  }
  return SQD_OK;
}

SQD_API int sqd_hdiag(sqd_ctx* c, double* out) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  DRAIN(c);
  SQD_HIP_CHECK(hipMemcpy(out, c->hdiag.p, (size_t)(c->row1 - c->row0) * c->nb * 8, hipMemcpyDeviceToHost));
  return SQD_OK;
}

SQD_API int sqd_init_guess(sqd_ctx* c, double* out) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  if (!out) return SQD_ERR_INVALID;
  SQD_TRY(c->io_out.reserve((size_t)c->D * 8));
  SQD_TRY(enqueue_init_guess(c, c->io_out.as<double>()));
  SQD_HIP_CHECK(hipMemcpyAsync(out, c->io_out.p, c->D * 8, hipMemcpyDeviceToHost, c->stream));
  SQD_STREAM_SYNC(c->stream);
  c->stage_pending = false;
  return SQD_OK;
}

// stage a host vector into tmp slot (X buffer is not used so a Davidson state is not disturbed)
static int upload_vec(sqd_ctx* c, const double* host, DevBuf& buf) {
  SQD_TRY(buf.reserve((size_t)c->D * 8));
  SQD_HIP_CHECK(hipMemcpyAsync(buf.p, host, c->D * 8, hipMemcpyHostToDevice, c->stream));
  return SQD_OK;
}

SQD_API int sqd_sigma(sqd_ctx* c, const double* cvec, double* sigma, int use_spin, double ss, double shift) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  if (!cvec || !sigma) return SQD_ERR_INVALID;
  DevBuf& in = c->io_in;
  DevBuf& out = c->io_out;
  SQD_TRY(upload_vec(c, cvec, in));
  SQD_TRY(out.reserve((size_t)c->D * 8));
  SQD_TRY(apply_h(c, in.as<double>(), out.as<double>(), use_spin, ss, shift));
  SQD_HIP_CHECK(hipMemcpyAsync(sigma, out.p, c->D * 8, hipMemcpyDeviceToHost, c->stream));
  SQD_STREAM_SYNC(c->stream);
  c->stage_pending = false;
  return SQD_OK;
}

SQD_API int sqd_contract_ss(sqd_ctx* c, const double* cvec, double* outv) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  if (!cvec || !outv) return SQD_ERR_INVALID;
  DevBuf& in = c->io_in;
  DevBuf& out = c->io_out;
  SQD_TRY(upload_vec(c, cvec, in));
  SQD_TRY(out.reserve((size_t)c->D * 8));
  SQD_TRY(launch_sigma(c, in.as<double>(), out.as<double>(), 1, false, 0.0, 0.0));
  SQD_HIP_CHECK(hipMemcpyAsync(outv, out.p, c->D * 8, hipMemcpyDeviceToHost, c->stream));
  SQD_STREAM_SYNC(c->stream);
  c->stage_pending = false;
  return SQD_OK;
}

SQD_API void sqd_davidson_default_opts(sqd_davidson_opts* o) {
  if (!o) return;
  o->tol = 1e-9;
  o->tol_residual = 0.0;
  o->lindep = 1e-14;
  o->max_cycle = 100;
  o->max_space = 12;
  o->use_spin = 0;
  o->ss = 0.0;
  o->shift = 0.2;
  o->verbose = 0;
  o->time_sigma_every = 0;
}

SQD_API int sqd_davidson(sqd_ctx* c, const sqd_davidson_opts* opts, const double* ci0, double* amps,
                         sqd_davidson_stats* stats) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  sqd_davidson_opts o;
  if (opts) o = *opts; else sqd_davidson_default_opts(&o);
  if (o.tol <= 0 || o.max_cycle < 1) {
    set_error("bad Davidson options");
    return SQD_ERR_INVALID;
  }
  c->want_timing = c->phase_timing || o.verbose;
  SQD_TRY(run_davidson(c, &o, ci0, stats));
  if (amps) {
    SQD_HIP_CHECK(hipMemcpyAsync(amps, c->sol.p, c->D * 8, hipMemcpyDeviceToHost, c->stream));
    SQD_STREAM_SYNC(c->stream);
  c->stage_pending = false;
  }
  return SQD_OK;
}

// resolve the state argument: host amps (uploaded to tmp2) or the resident solution
static int state_ptr(sqd_ctx* c, const double* amps, const double** d) {
  if (amps) {
    SQD_TRY(upload_vec(c, amps, c->io_in));
    *d = c->io_in.as<double>();
    return SQD_OK;
  }
  if (!c->have_solution) {
    set_error("no resident solution: run sqd_davidson or pass amplitudes");
    return SQD_ERR_STATE;
  }
  *d = c->sol.as<double>();
  return SQD_OK;
}

static int expectation(sqd_ctx* c, const double* amps, int mode, double* outv) {
  const double* d = nullptr;
  SQD_TRY(state_ptr(c, amps, &d));
  SQD_TRY(c->tmp1.reserve((size_t)c->D * 8));
  SQD_TRY(launch_sigma(c, d, c->tmp1.as<double>(), mode, false, 0.0, 0.0));
  double num = 0.0, den = 0.0;
  SQD_TRY(dev_dot(c, d, c->tmp1.as<double>(), &num));
  SQD_TRY(dev_dot(c, d, d, &den));
  if (!(den > 0.0)) {
    set_error("state has zero norm");
    return SQD_ERR_INVALID;
  }
  *outv = num / den;
  return SQD_OK;
}

SQD_API int sqd_observables(sqd_ctx* c, const double* amps, double* e, double* s2, double* occ_a, double* occ_b) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  const double* d = nullptr;
  SQD_TRY(state_ptr(c, amps, &d));
  std::vector<double> out(4 + 2 * c->norb);
  SQD_TRY(dev_observables(c, d, out.data()));
  if (!(out[2] > 0.0)) {
    set_error("state has zero norm");
    return SQD_ERR_INVALID;
  }
  if (e) *e = out[0] / out[2];
  if (s2) *s2 = out[1] / out[2];
  for (int p = 0; p < c->norb; ++p) {
    if (occ_a) occ_a[p] = out[3 + p] / out[2];
    if (occ_b) occ_b[p] = out[3 + c->norb + p] / out[2];
  }
  return SQD_OK;
}

// after the Davidson and the observables kernel of a solve have completed: the run's statistics and the results
// derived from the reduced records (energy: see sqd_solve)
static int solve_collect(sqd_ctx* c, const sqd_davidson_opts& o, int form, sqd_davidson_stats* stats, double* e, double* s2,
                         double* occ_a, double* occ_b) {
  sqd_davidson_stats local;
  sqd_davidson_stats* stp = stats ? stats : &local;
  SQD_TRY(davidson_collect(c, stp));
  std::vector<double> out(4 + 2 * c->norb);
  dev_observables_collect(c, out.data());
  const double cc = out[2];
  if (!(cc > 0.0)) {
    set_error("state has zero norm");
    return SQD_ERR_INVALID;
  }
  const double ct = out[1] / cc, tt = out[3 + 2 * c->norb] / cc;
  double penalty = 0.0;
  if (form == 1) penalty = ct - o.ss;
  else if (form == 2) penalty = tt - 2.0 * o.ss * ct + o.ss * o.ss;
  if (e) *e = stp->e_davidson - (form ? o.shift * penalty : 0.0);
  if (s2) *s2 = ct;
  for (int p = 0; p < c->norb; ++p) {
    if (occ_a) occ_a[p] = out[3 + p] / cc;
    if (occ_b) occ_b[p] = out[3 + c->norb + p] / cc;
  }
  return SQD_OK;
}
static int penalty_form(const sqd_ctx* c, const sqd_davidson_opts& o) {
  int form = o.use_spin;
  if (form == 3) {
    const double szh = 0.5 * std::abs(c->nelec[0] - c->nelec[1]);
    form = (o.ss < szh * (szh + 1.0) + 0.1) ? 1 : 2;
  }
  return form;
}

// One call for the whole of solve_fermion's device work: Davidson, then the observables' kernels on the
// compute stream WHILE the amplitudes travel to the host on the copy stream; one synchronisation.
//
// Energy: the reference recomputes <c|H|c> from the returned state because pyscf's eigenvalue includes the spin
// penalty (fermion.py:717-732, :825-827).  The Ritz value IS the Rayleigh quotient of the returned vector with
// the operator that was iterated (every A X_v in the basis is an exact sigma build), so
//     <c|H|c> = e_davidson - shift * <penalty>,   <penalty> = <S^2> - ss   or   <(S^2 - ss)^2> = |S^2 c - ss c|^2,
// both available from the one S^2 c that <S^2> needs anyway: no extra H sigma build.  Without a penalty and with
// s2 == NULL (the sci_solver seam never reads <S^2>, fermion.py:684-742) no sigma build follows the Davidson at all.
SQD_API int sqd_solve(sqd_ctx* c, const sqd_davidson_opts* opts, const double* ci0, double* amps,
                      sqd_davidson_stats* stats, double* e, double* s2, double* occ_a, double* occ_b) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  sqd_davidson_opts o;
  if (opts) o = *opts; else sqd_davidson_default_opts(&o);
  if (o.tol <= 0 || o.max_cycle < 1) {
    set_error("bad Davidson options");
    return SQD_ERR_INVALID;
  }
  c->want_timing = c->phase_timing || o.verbose;
  const size_t bytes = (size_t)c->D * 8;
  double* twin = nullptr;
  if (amps && bytes <= (size_t(64) << 20)) twin = static_cast<double*>(pinned_device_ptr(amps, bytes));
  const bool late = c->async_state && twin != nullptr && !c->parent;
  if (late) {
    // the previous solve's state may still be leaving `sol` (k_state_copy on the copy stream): this solve forms its
    // solution in the other buffer.  What last left THAT one -- two solves ago -- has landed long since; checked, not assumed.
    if (c->sol_alt_ticket > 0 && !state_copy_landed(c, c->sol_alt_ticket)) SQD_TRY(state_copy_wait(c, c->sol_alt_ticket));
    std::swap(c->sol, c->sol_alt);
    std::swap(c->sol_ticket, c->sol_alt_ticket);
    c->have_solution = false;
  }
  SQD_TRY(run_davidson(c, &o, ci0, nullptr, /*defer_sync=*/true));
  // Three ways for the state to reach the caller:
  //  * the caller's buffer is page-locked (sqd_host_alloc) and the state is small: the observables' kernel, which
  //    reads every element anyway, also WRITES it there (posted PCIe writes) -- no second stream, no event, no DMA
  //    set-up, one wait.  (Beyond 64 MB the DMA engine on the copy stream wins: it overlaps the S^2 sigma build.)
  //  * small and pageable: through the context's pinned staging buffer on the copy stream (a truly asynchronous copy)
  //  * large: DMA straight to the caller's memory on the copy stream
  const bool by_copy = amps && !twin;
  const bool staged = by_copy && bytes <= (size_t(64) << 20);
  if (by_copy) {
    SQD_HIP_CHECK(hipEventRecord(c->ev_sol, c->stream));
    SQD_HIP_CHECK(hipStreamWaitEvent(c->copy_stream, c->ev_sol, 0));
    if (staged) {
      if (c->h_amps_cap < bytes) {
        if (c->h_amps) SQD_HIP_CHECK(hipHostFree(c->h_amps));
        c->h_amps = nullptr;
        c->h_amps_cap = 0;
        SQD_HIP_CHECK(hipHostMalloc((void**)&c->h_amps, bytes, hipHostMallocDefault));
        c->h_amps_cap = bytes;
      }
      SQD_HIP_CHECK(hipMemcpyAsync(c->h_amps, c->sol.p, bytes, hipMemcpyDeviceToHost, c->copy_stream));
    }
  }
  int form = o.use_spin;
  if (form == 3) {
    const double szh = 0.5 * std::abs(c->nelec[0] - c->nelec[1]);
    form = (o.ss < szh * (szh + 1.0) + 0.1) ? 1 : 2;
  }
  const bool need_s2 = (s2 != nullptr) || form != 0;
  // sqd_ctx_set_async_state: the state follows the results (k_state_copy on the copy stream); the call returns with the
  // results and a ticket, sqd_ctx_state_wait(ticket) says when the caller's buffer is complete
  SQD_TRY(dev_observables_enqueue(c, c->sol.as<double>(), /*with_h=*/false, /*with_s2=*/need_s2, twin, late));
  // (the copy's start event sits BEHIND the observables' kernel on purpose: enqueued in front of it -- 19 us of posted PCIe
  // writes beside the kernel's 15 instead of after them -- a headline call takes 0.154 ms instead of 0.132, measured in
  // round 6, profiles/r06/kernel_boundaries_probe.txt)
  long long ticket = 0;
  if (late) {
    SQD_TRY(state_copy_enqueue(c, twin, &ticket));
    c->sol_ticket = ticket;
  }
  if (c->enqueue_hook) c->enqueue_hook(c->enqueue_hook_user);  // (the caller's collective, right behind the last kernel)
  if (by_copy && !staged)
    SQD_HIP_CHECK(hipMemcpyAsync(amps, c->sol.p, bytes, hipMemcpyDeviceToHost, c->copy_stream));
  if (by_copy) SQD_STREAM_SYNC(c->copy_stream);
  if (staged) std::memcpy(amps, c->h_amps, bytes);
  int rc = dev_observables_wait(c, /*whole_kernel=*/!late);
  if (rc == SQD_OK) {
    c->stage_pending = false;  // (the results are there: every upload in front of them in the stream has been consumed)
    rc = solve_collect(c, o, form, stats, e, s2, occ_a, occ_b);
  }
  // a call that fails hands no ticket back: the caller will recycle its page-locked block at once, so nothing may
  // still be writing into it (k_state_copy on the copy stream)
  if (rc != SQD_OK && late && ticket > 0) state_copy_wait(c, ticket);
  if (stats) stats->state_ticket = (rc == SQD_OK) ? ticket : 0;
  return rc;
}

// sqd_set_subspace + sqd_solve in ONE crossing of the boundary: the body of reference solve_fermion / solve_sci from
// the formatted CI strings to the returned tuple (fermion.py:797-830, :713-742).  Between two native calls the
// Python side spends ~20 us (ctypes marshalling, the dims query), a tenth of a whole headline-size solve.
SQD_API int sqd_solve_strings(sqd_ctx* c, const uint64_t* strs_a, int64_t na, const uint64_t* strs_b, int64_t nb,
                              const sqd_davidson_opts* opts, const double* ci0, double* amps, sqd_davidson_stats* stats,
                              double* e, double* s2, double* occ_a, double* occ_b, int* nelec_a, int* nelec_b) {
  CTX_ENTER(c);
  c->want_timing = c->phase_timing || (opts && opts->verbose);
  SQD_TRY(build_subspace(c, strs_a, na, strs_b, nb));
  if (nelec_a) *nelec_a = c->nelec[0];
  if (nelec_b) *nelec_b = c->nelec[1];
  return sqd_solve(c, opts, ci0, amps, stats, e, s2, occ_a, occ_b);
}

// ---- batched solve of a whole ci_strings list (declared and documented in include/sqd_hip.h)
static int sub_ctx_create(sqd_ctx* parent, sqd_ctx** out) {
  sqd_ctx* c = new sqd_ctx();
  c->device = parent->device;
  c->norb = parent->norb;
  c->nnorb = parent->nnorb;
  c->num_cu = parent->num_cu;
  c->lds_bytes = parent->lds_bytes;
  c->stream = parent->stream;
  c->owns_stream = false;
  c->parent = parent;
  c->h1.set_view(parent->h1.p);
  c->eri4.set_view(parent->eri4.p);
  c->eri_pp.set_view(parent->eri_pp.p);
  c->jm.set_view(parent->jm.p);
  c->km.set_view(parent->km.p);
  c->jdiag.set_view(parent->jdiag.p);
  hipError_t e = hipStreamCreate(&c->copy_stream);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_sol);
  if (e == hipSuccess) e = hipEventCreate(&c->ev_aux);
  for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreate(&c->ev[i]);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_pinned, 4096 * sizeof(double), hipHostMallocDefault);
  if (e == hipSuccess) e = hipHostMalloc((void**)&c->h_mail, 1024 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&c->d_mail, c->h_mail, 0);
  if (e != hipSuccess) {
    set_error(std::string("sub-context resources: ") + hipGetErrorString(e));
    sqd_ctx_destroy(c);
    return SQD_ERR_HIP;
  }
  std::memset(c->h_mail, 0, 1024 * sizeof(double));
  *out = c;
  return SQD_OK;
}

SQD_API int sqd_solve_batch(sqd_ctx* c, int nbatch, const uint64_t* const* strs_a, const int64_t* na,
                            const uint64_t* const* strs_b, const int64_t* nb, const sqd_davidson_opts* opts,
                            double* const* amps, double* best_amps, int* best, sqd_davidson_stats* stats, double* e,
                            double* s2, double* occ_a, double* occ_b, int* nelec_a, int* nelec_b) {
  bool hook_done = false;  // the caller's enqueue hook fires once per call, behind the last launch
  CTX_ENTER(c);
  if (c->parent) {
    set_error("sqd_solve_batch on a sub-context");
    return SQD_ERR_STATE;
  }
  if (nbatch < 1 || !strs_a || !na || !strs_b || !nb) {
    set_error("sqd_solve_batch: null argument or empty batch list");
    return SQD_ERR_INVALID;
  }
  if ((best || best_amps) && !e) {
    set_error("sqd_solve_batch: `e` is required with best / best_amps");
    return SQD_ERR_INVALID;
  }
  sqd_davidson_opts o;
  if (opts) o = *opts; else sqd_davidson_default_opts(&o);
  if (o.tol <= 0 || o.max_cycle < 1) {
    set_error("bad Davidson options");
    return SQD_ERR_INVALID;
  }
  o.verbose = 0;
  o.time_sigma_every = 0;
  while ((int)c->subs.size() < nbatch) {
    sqd_ctx* sub = nullptr;
    SQD_TRY(sub_ctx_create(c, &sub));
    c->subs.push_back(sub);
  }
  std::vector<sqd_ctx*> subs(c->subs.begin(), c->subs.begin() + nbatch);
  // The latest solutions become the previous ones.  A call that FAILS further down (a bad string list, a zero-norm
  // state, a HIP error) rotates them back: the caller's bookkeeping of which call's states are resident (generation
  // counter of the Python layer) moves only with calls that succeed.
  struct Rotation {
    sqd_ctx* c;
    int batch_n;
    std::vector<int64_t> d_prev, d_now;
    std::vector<char> had_solution;
    bool keep = false;
    ~Rotation() {
      if (keep) return;
      for (size_t p = 0; p < c->subs.size() && p < d_prev.size(); ++p) {
        sqd_ctx* sub = c->subs[p];
        std::swap(sub->sol, sub->sol_prev);
        sub->D_prev = d_prev[p];
        sub->D = d_now[p];  // (the size of the solution that is resident again; the tables may be the failed call's)
        sub->have_solution = had_solution[p] != 0;
      }
      c->batch_n = batch_n;
    }
  } rotation{c, c->batch_n, {}, {}, {}};
  c->batch_n = 0;
  for (size_t p = 0; p < c->subs.size(); ++p) {
    sqd_ctx* sub = c->subs[p];
    rotation.d_prev.push_back(sub->D_prev);
    rotation.d_now.push_back(sub->D);
    rotation.had_solution.push_back(sub->have_solution ? 1 : 0);
    const bool had = (int)p < c->batch_n_prev_valid && sub->have_solution;
    std::swap(sub->sol, sub->sol_prev);
    sub->D_prev = had ? sub->D : 0;
    sub->have_solution = false;
  }
  for (int p = 0; p < nbatch; ++p) {
    sqd_ctx* sub = subs[p];
    sub->stream = c->stream;
    sub->want_timing = false;
    sub->record_out = c->record_out ? c->record_out + (int64_t)p * c->record_stride : nullptr;
    sub->record_stride = 0;
  }
  hipStream_t st = c->stream;
  // ---- tables of every subspace: 2 copies + 4-5 launches for the whole batch
  SQD_TRY(build_subspace_batch(c, subs, strs_a, na, strs_b, nb));
  for (int p = 0; p < nbatch; ++p) {
    if (nelec_a) nelec_a[p] = subs[p]->nelec[0];
    if (nelec_b) nelec_b[p] = subs[p]->nelec[1];
  }
  // ---- which subspaces the batched launches cover; the others are solved one by one at the end
  std::vector<int> bidx, sidx;
  for (int p = 0; p < nbatch; ++p) {
    if (sigma_batch_supported(subs[p]) && penalty_form(subs[p], o) != 2) bidx.push_back(p);
    else sidx.push_back(p);
  }
  bool need_s2 = (s2 != nullptr);
  for (int p : bidx) need_s2 = need_s2 || penalty_form(subs[p], o) != 0;
  if (!bidx.empty()) {
    std::vector<sqd_ctx*> bs;
    std::vector<double*> twins;
    for (int p : bidx) {
      bs.push_back(subs[p]);
      const size_t bytes = (size_t)subs[p]->D * 8;
      double* twin = nullptr;
      if (amps && amps[p] && bytes <= (size_t(64) << 20)) twin = static_cast<double*>(pinned_device_ptr(amps[p], bytes));
      twins.push_back(twin);
    }
    // ---- the solver's argument records (Davidson, sigma, observables) of every subspace: one copy
    BatchStage& s3 = c->bstage[2];
    SQD_TRY(s3.reserve(davidson_batch_bytes(bs.size()) + observables_batch_bytes(bs.size())));
    size_t off = 0;
    DavBatchPlan dplan;
    ObsBatchPlan oplan;
    char* d3 = static_cast<char*>(s3.dev.p);
    SQD_TRY(davidson_batch_prepare(c, bs, &o, s3.host, d3, &off, &dplan));
    SQD_TRY(observables_batch_prepare(c, bs, need_s2, twins, s3.host, d3, &off, &oplan));
    SQD_HIP_CHECK(hipMemcpyAsync(d3, s3.host, off, hipMemcpyHostToDevice, st));
    SQD_TRY(davidson_batch_run(c, bs, dplan));
    SQD_TRY(observables_batch_launch(c, oplan));
    // states whose buffer the observables kernel could not write itself (pageable memory, or beyond 64 MB)
    for (size_t k = 0; k < bidx.size(); ++k) {
      const int p = bidx[k];
      if (amps && amps[p] && !twins[k])
        SQD_HIP_CHECK(hipMemcpyAsync(amps[p], subs[p]->sol.p, (size_t)subs[p]->D * 8, hipMemcpyDeviceToHost, st));
    }
    if (c->enqueue_hook && sidx.empty()) {  // (every record of the call is on its way: the caller's collective goes here)
      c->enqueue_hook(c->enqueue_hook_user);
      hook_done = true;
    }
    for (sqd_ctx* sub : bs) SQD_TRY(spin_wait_word(sub->h_mail + 3 * 128 + 200, sub->obs_seq, st));
    SQD_TRY(spin_stream_sync(st));
    for (int p : bidx) {
      sqd_ctx* sub = subs[p];
      sub->stage_pending = false;
      SQD_TRY(solve_collect(sub, o, penalty_form(sub, o), stats ? &stats[p] : nullptr, e ? &e[p] : nullptr,
                            s2 ? &s2[p] : nullptr, occ_a ? occ_a + (size_t)p * c->norb : nullptr,
                            occ_b ? occ_b + (size_t)p * c->norb : nullptr));
    }
  }
  for (int p : sidx) {
    double e_p = 0.0;
    SQD_TRY(sqd_solve(subs[p], &o, nullptr, amps ? amps[p] : nullptr, stats ? &stats[p] : nullptr, &e_p,
                      s2 ? &s2[p] : nullptr, occ_a ? occ_a + (size_t)p * c->norb : nullptr,
                      occ_b ? occ_b + (size_t)p * c->norb : nullptr));
    if (e) e[p] = e_p;
  }
  if (c->enqueue_hook && !hook_done) c->enqueue_hook(c->enqueue_hook_user);
  c->batch_n = nbatch;
  c->batch_n_prev_valid = nbatch;
  rotation.keep = true;
  if (best || best_amps) {
    int w = 0;
    for (int p = 1; p < nbatch; ++p)
      if (e[p] < e[w]) w = p;  // first minimum, as numpy.argmin (reference fermion.py:577)
    if (best) *best = w;
    if (best_amps) {
      const size_t bytes = (size_t)subs[w]->D * 8;
      if (amps && amps[w]) std::memcpy(best_amps, amps[w], bytes);
      else {
        SQD_HIP_CHECK(hipMemcpyAsync(best_amps, subs[w]->sol.p, bytes, hipMemcpyDeviceToHost, st));
        SQD_TRY(spin_stream_sync(st));
      }
    }
  }
  return SQD_OK;
}

SQD_API int sqd_batch_ctx(sqd_ctx* c, int index, sqd_ctx** sub) {
  CTX_ENTER(c);
  if (!sub || index < 0 || index >= c->batch_n) {
    set_error("sqd_batch_ctx: no such batch in the latest sqd_solve_batch");
    return SQD_ERR_INVALID;
  }
  *sub = c->subs[index];
  return SQD_OK;
}

SQD_API int sqd_batch_state(sqd_ctx* c, int index, int age, double* amps) {
  CTX_ENTER(c);
  if (!amps || index < 0 || index >= (int)c->subs.size() || age < 0 || age > 1) {
    set_error("sqd_batch_state: no such batch");
    return SQD_ERR_INVALID;
  }
  sqd_ctx* sub = c->subs[index];
  const void* src = nullptr;
  size_t bytes = 0;
  if (age == 0 && index < c->batch_n && sub->have_solution) {
    src = sub->sol.p;
    bytes = (size_t)sub->D * 8;
  } else if (age == 1 && sub->D_prev > 0 && sub->sol_prev.p) {
    src = sub->sol_prev.p;
    bytes = (size_t)sub->D_prev * 8;
  }
  if (!src) {
    set_error("sqd_batch_state: that solution is no longer resident");
    return SQD_ERR_STATE;
  }
  SQD_HIP_CHECK(hipMemcpyAsync(amps, src, bytes, hipMemcpyDeviceToHost, c->stream));
  SQD_TRY(spin_stream_sync(c->stream));
  return SQD_OK;
}

SQD_API int sqd_energy(sqd_ctx* c, const double* amps, double* e) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  return expectation(c, amps, 0, e);
}
SQD_API int sqd_spin_square(sqd_ctx* c, const double* amps, double* s2) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  return expectation(c, amps, 1, s2);
}
SQD_API int sqd_rdm1s(sqd_ctx* c, const double* amps, double* dm1a, double* dm1b) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  const double* d = nullptr;
  SQD_TRY(state_ptr(c, amps, &d));
  return dev_rdm1s(c, d, dm1a, dm1b);
}
SQD_API int sqd_rdm2(sqd_ctx* c, const double* amps, double* dm2) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  const double* d = nullptr;
  SQD_TRY(state_ptr(c, amps, &d));
  return dev_rdm2(c, d, dm2);
}

SQD_API int sqd_rdm2s(sqd_ctx* c, const double* amps, double* dm2aa, double* dm2ab, double* dm2bb) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  if (!dm2aa || !dm2ab || !dm2bb) return SQD_ERR_INVALID;
  const double* d = nullptr;
  SQD_TRY(state_ptr(c, amps, &d));
  return dev_rdm2s(c, d, dm2aa, dm2ab, dm2bb);
}

SQD_API int sqd_time_sigma(sqd_ctx* c, int reps, int use_spin, double ss, double shift, double* ms_per_sigma) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  if (reps < 1 || !ms_per_sigma) return SQD_ERR_INVALID;
  const double* d = nullptr;
  if (c->have_solution) {
    d = c->sol.as<double>();
  } else {
    SQD_TRY(sol_writer_guard(c));
    SQD_TRY(c->sol.reserve((size_t)c->D * 8));
    SQD_HIP_CHECK(hipMemsetAsync(c->sol.p, 0, c->D * 8, c->stream));
    d = c->sol.as<double>();
  }
  SQD_TRY(c->tmp1.reserve((size_t)c->D * 8));
  if (use_spin == 2 || use_spin == 3) SQD_TRY(c->tmp2.reserve((size_t)c->D * 8));
  SQD_TRY(apply_h(c, d, c->tmp1.as<double>(), use_spin == 2 ? 0 : use_spin, ss, shift));  // warm-up
  SQD_HIP_CHECK(hipEventRecord(c->ev[2], c->stream));
  for (int i = 0; i < reps; ++i) SQD_TRY(apply_h(c, d, c->tmp1.as<double>(), use_spin == 2 ? 0 : use_spin, ss, shift));
  SQD_HIP_CHECK(hipEventRecord(c->ev[3], c->stream));
  SQD_STREAM_SYNC(c->stream);
  c->stage_pending = false;
  float ms = 0.f;
  SQD_HIP_CHECK(hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
  *ms_per_sigma = (double)ms / reps;
  return SQD_OK;
}

// ... with a HIP-event bracket around EVERY launch of the dominant sigma kernel and an empty bracket right behind it:
// out[0] / out[1] = mean / median kernel bracket, out[2] / out[3] = mean / median empty bracket (ms).  `reps` samples of a
// 6 us kernel average the jitter of a 5 us event record away; bench.py's roofline leg runs it BEHIND the timed region.
SQD_API int sqd_time_sigma_brackets(sqd_ctx* c, int reps, int use_spin, double ss, double shift, double* out4) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  if (reps < 1 || reps > 4096 || !out4) return SQD_ERR_INVALID;
  const double* d = nullptr;
  if (c->have_solution) {
    d = c->sol.as<double>();
  } else {
    SQD_TRY(sol_writer_guard(c));
    SQD_TRY(c->sol.reserve((size_t)c->D * 8));
    SQD_HIP_CHECK(hipMemsetAsync(c->sol.p, 0, c->D * 8, c->stream));
    d = c->sol.as<double>();
  }
  SQD_TRY(c->tmp1.reserve((size_t)c->D * 8));
  std::vector<hipEvent_t> ev((size_t)3 * reps, nullptr);
  int rc = SQD_OK;
  auto cleanup = [&]() {
    for (hipEvent_t e : ev)
      if (e) (void)hipEventDestroy(e);
  };
  for (hipEvent_t& e : ev)
    if (hipEventCreate(&e) != hipSuccess) {
      cleanup();
      set_error("sqd_time_sigma_brackets: hipEventCreate failed");
      return SQD_ERR_HIP;
    }
  rc = apply_h(c, d, c->tmp1.as<double>(), use_spin, ss, shift);  // warm-up
  for (int i = 0; i < reps && rc == SQD_OK; ++i) {
    if (hipEventRecord(ev[3 * i], c->stream) != hipSuccess) rc = SQD_ERR_HIP;
    c->ev_after_sigma_kernel = ev[3 * i + 1];  // recorded by the launcher right behind the dominant kernel
    if (rc == SQD_OK) rc = apply_h(c, d, c->tmp1.as<double>(), use_spin, ss, shift);
    if (c->ev_after_sigma_kernel) {  // (a formulation that does not record it: bracket the whole application)
      c->ev_after_sigma_kernel = nullptr;
      if (rc == SQD_OK && hipEventRecord(ev[3 * i + 1], c->stream) != hipSuccess) rc = SQD_ERR_HIP;
    }
    if (rc == SQD_OK && hipEventRecord(ev[3 * i + 2], c->stream) != hipSuccess) rc = SQD_ERR_HIP;
  }
  if (rc == SQD_OK) rc = spin_stream_sync(c->stream);
  c->stage_pending = false;
  if (rc == SQD_OK) {
    std::vector<double> k((size_t)reps), e((size_t)reps);
    for (int i = 0; i < reps; ++i) {
      float a = 0.f, b = 0.f;
      if (hipEventElapsedTime(&a, ev[3 * i], ev[3 * i + 1]) != hipSuccess || hipEventElapsedTime(&b, ev[3 * i + 1], ev[3 * i + 2]) != hipSuccess) {
        rc = SQD_ERR_HIP;
        break;
      }
      k[i] = a;
      e[i] = b;
    }
    if (rc == SQD_OK) {
      auto mean = [](const std::vector<double>& v) { double s = 0; for (double x : v) s += x; return s / (double)v.size(); };
      auto median = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
      out4[0] = mean(k);
      out4[1] = median(k);
      out4[2] = mean(e);
      out4[3] = median(e);
    }
  }
  cleanup();
  if (rc == SQD_ERR_HIP) set_error("sqd_time_sigma_brackets: HIP event failure");
  return rc;
}

SQD_API int sqd_time_dense(sqd_ctx* c, int reps, int copies, double* ms_per_launch, double* flops_per_launch) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  NEED_ALL_ROWS(c);
  if (reps < 1 || !ms_per_launch || !flops_per_launch) return SQD_ERR_INVALID;
  if (!c->have_solution) {
    SQD_TRY(sol_writer_guard(c));
    SQD_TRY(c->sol.reserve((size_t)c->D * 8));
    SQD_HIP_CHECK(hipMemsetAsync(c->sol.p, 0, c->D * 8, c->stream));
  }
  return time_dense_product(c, c->sol.as<double>(), reps, copies, ms_per_launch, flops_per_launch);
}

SQD_API int sqd_sigma_bytes(sqd_ctx* c, double* bytes) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  if (!bytes) return SQD_ERR_INVALID;
  // B_sigma = 8 D (read c) + 8 D (write sigma) + 8 * populated links + 8 (nnorb_s^2 + nnorb_a^2)
  const double D = (double)c->D;
  const double links = (double)(c->sp[0].n_s + c->sp[0].n_d + c->sp[1].n_s + c->sp[1].n_d) +
                       (double)c->na * c->nelec[0] + (double)c->nb * c->nelec[1];
  const double ns = (double)c->nnorb, nas = (double)c->norb * (c->norb - 1) / 2.0;
  *bytes = 16.0 * D + 8.0 * links + 8.0 * (ns * ns + nas * nas);
  return SQD_OK;
}

SQD_API int sqd_sigma_kernel(sqd_ctx* c, int* kind, int* rows_per_workgroup) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  if (kind) *kind = c->sig_lists ? 4 : c->sig_rows > 0 ? 2 : (c->sig_direct ? 1 : (c->sig_opp ? (c->opp_src ? 7 : 6) : (c->sig_spmm ? 5 : (c->sig_dense ? 3 : 0))));
  if (rows_per_workgroup) *rows_per_workgroup = c->sig_spmm ? spmm_rows_per_group(c) : c->sig_rows;
  return SQD_OK;
}

SQD_API int sqd_sigma_bytes_needed(sqd_ctx* c, double* bytes) {
  CTX_ENTER(c);
  NEED_SUBSPACE(c);
  if (!bytes) return SQD_ERR_INVALID;
  // what THIS formulation must move once per sigma (every object at its stored width, each counted once):
  // c, sigma, hdiag; the beta link records (singles 16 B, doubles 12 B) and the alpha ones (merged list 12 B +
  // 8 B per single); one packed integral row and one J_beta row per distinct orbital pair among the alpha
  // singles (at most min(links, nnorb)); the J_alpha row of every alpha string
  const double D = (double)c->D, nn = (double)c->nnorb;
  const double nsa = (double)c->sp[0].n_s, nda = (double)c->sp[0].n_d, nsb = (double)c->sp[1].n_s, ndb = (double)c->sp[1].n_d;
  const double pairs = nsa < nn ? nsa : nn;
  *bytes = 24.0 * D + 16.0 * nsb + 12.0 * ndb + 12.0 * (nsa + nda) + 8.0 * nsa + pairs * 8.0 * (nn + (double)c->nb) +
           (nsb > 0 ? 8.0 * nn * (double)c->na : 0.0);
  return SQD_OK;
}

