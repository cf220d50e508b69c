// Host-side kernel-LOGIC emulator for the `-m "not gpu"` unit tests.  TEST TOOL ONLY.
//
// It lets the unmodified HIP kernel sources under qiskit-addon-sqd_amd/csrc be
// compiled with g++ and executed on the host (one fiber per GPU thread), so index /
// sign / bounds bugs are caught in the GPU-less authoring container before a
// gpurun call is spent.  It is NOT a backend: the product loader
// (qiskit_addon_sqd_amd/_capi.py) only ever loads the hipcc-built
// libsqd_hip.so, and tests/emu builds into tests/emu/_build/libsqd_emu.so which
// nothing outside tests/ knows about.
//
// Model: one fiber per GPU thread of a block, all on the launching OS thread;
// the blocks of a grid run one after another; wave = 64 consecutive threads;
// wave-level builtins rendezvous on a per-wave barrier, __syncthreads on a
// per-block barrier.  "Device memory" is host memory.  Streams and events are
// no-ops.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <functional>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint2 { unsigned x, y; };
struct int2 { int x, y; };
struct uint4 { unsigned x, y, z, w; };
struct double2 { double x, y; };
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline double2 make_double2(double a, double b) { return double2{a, b}; }

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum { hipDeviceAttributeMultiprocessorCount = 1, hipDeviceAttributeMaxSharedMemoryPerBlock = 2 };
enum { hipHostMallocDefault = 0, hipHostMallocPortable = 1, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(emu::dyn_smem());

namespace emu {
// Execution model: the threads of a block are FIBERS (user-level contexts with their own stacks) multiplexed on the
// launching OS thread; a barrier is "mark myself waiting and switch to the scheduler".  (The first version ran one OS
// thread per GPU thread on std::barrier: with 512-thread blocks on an 8-core box the futex traffic was > 90 % of the
// CPU suite's run time.)  Threads that have returned from the kernel count as arrived at every later barrier, as on
// the hardware.
struct Wave {
  unsigned long long slot[64];
};
struct Block {
  std::vector<Wave> waves;
  std::vector<char> dyn;
};
inline Block*& cur_block() { static Block* b = nullptr; return b; }
struct TL { dim3 tid, bid, bdim, gdim; };
enum { FIBER_RUNNABLE = 0, FIBER_AT_BLOCK_BARRIER = 1, FIBER_AT_WAVE_BARRIER = 2, FIBER_DONE = 3 };
struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  int state = FIBER_DONE;
  TL tl;
};
constexpr size_t FIBER_STACK = 256 * 1024;
struct Sched {  // one for the process; guarded by launch_mu
  std::vector<Fiber> fib;
  void* main_sp = nullptr;
  int cur = 0;
  std::function<void()> body;
};
inline Sched& sched() { static Sched* s = new Sched(); return *s; }
inline TL& tl() { Sched& s = sched(); return s.fib[s.cur].tl; }
inline void* dyn_smem() {
  // 16-byte aligned start
  auto p = reinterpret_cast<uintptr_t>(cur_block()->dyn.data());
  return reinterpret_cast<void*>((p + 15) & ~uintptr_t(15));
}
inline Wave& my_wave() { return cur_block()->waves[tl().tid.x / 64]; }
inline std::mutex& atomic_mu() { static std::mutex m; return m; }
inline std::mutex& launch_mu() { static std::mutex m; return m; }

#if !defined(__x86_64__)
#error "tests/emu: the fiber switch is written for x86-64"
#endif
// save the callee-saved registers and the stack pointer of the running context in *save_sp, continue on load_sp
__attribute__((naked, noinline, unused)) static void emu_switch(void** /*save_sp*/, void* /*load_sp*/) {
  __asm__ volatile(
      "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
      "movq %rsp, (%rdi)\n\tmovq %rsi, %rsp\n\t"
      "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\tret\n\t");
}
// the running fiber gives way to the scheduler (state = why)
inline void fiber_yield(int state) {
  Sched& s = sched();
  Fiber& f = s.fib[s.cur];
  f.state = state;
  emu_switch(&f.sp, s.main_sp);
}
__attribute__((unused)) static void fiber_main() {
  Sched& s = sched();
  s.body();
  fiber_yield(FIBER_DONE);
  std::abort();  // a finished fiber is never resumed
}
inline void fiber_prepare(Fiber& f) {
  if (!f.stack) f.stack = static_cast<char*>(std::malloc(FIBER_STACK));
  auto top = reinterpret_cast<uintptr_t>(f.stack + FIBER_STACK) & ~uintptr_t(15);
  void** sp = reinterpret_cast<void**>(top);
  *--sp = nullptr;                                   // keeps rsp = 8 (mod 16) at the entry of fiber_main, as after a call
  *--sp = reinterpret_cast<void*>(&fiber_main);      // "return address" of the first switch
  for (int r = 0; r < 6; ++r) *--sp = nullptr;       // rbp rbx r12 r13 r14 r15
  f.sp = sp;
  f.state = FIBER_RUNNABLE;
}
// run the T fibers of one block to completion
inline void run_block(int T) {
  Sched& s = sched();
  int remaining = T;
  const int nw = (T + 63) / 64;
  while (remaining > 0) {
    bool progress = false;
    for (int t = 0; t < T; ++t) {
      if (s.fib[t].state != FIBER_RUNNABLE) continue;
      s.cur = t;
      emu_switch(&s.main_sp, s.fib[t].sp);
      progress = true;
      if (s.fib[t].state == FIBER_DONE) --remaining;
    }
    // barriers whose every live participant has arrived open
    for (int w = 0; w < nw; ++w) {
      const int lo = 64 * w, hi = std::min(T, lo + 64);
      int waiting = 0, other = 0;
      for (int t = lo; t < hi; ++t) {
        waiting += s.fib[t].state == FIBER_AT_WAVE_BARRIER;
        other += s.fib[t].state == FIBER_RUNNABLE || s.fib[t].state == FIBER_AT_BLOCK_BARRIER;
      }
      if (waiting && !other) {
        for (int t = lo; t < hi; ++t)
          if (s.fib[t].state == FIBER_AT_WAVE_BARRIER) s.fib[t].state = FIBER_RUNNABLE;
        progress = true;
      }
    }
    int at_block = 0, elsewhere = 0;
    for (int t = 0; t < T; ++t) {
      at_block += s.fib[t].state == FIBER_AT_BLOCK_BARRIER;
      elsewhere += s.fib[t].state == FIBER_RUNNABLE || s.fib[t].state == FIBER_AT_WAVE_BARRIER;
    }
    if (at_block && !elsewhere) {
      for (int t = 0; t < T; ++t)
        if (s.fib[t].state == FIBER_AT_BLOCK_BARRIER) s.fib[t].state = FIBER_RUNNABLE;
      progress = true;
    }
    if (!progress) {
      std::fprintf(stderr, "tests/emu: barrier deadlock (divergent __syncthreads / wave builtin)\n");
      std::abort();
    }
  }
}

template <class K, class... A>
void launch(K kernel, dim3 grid, dim3 block, size_t shmem, A... args) {
  // one kernel at a time: the scheduler, the current block and the `__shared__ static` variables are process-wide,
  // and host threads (solve_sci_batch runs several contexts concurrently) may launch at the same moment
  std::lock_guard<std::mutex> launch_guard(launch_mu());
  const int T = block.x;
  Sched& s = sched();
  if ((int)s.fib.size() < T) s.fib.resize(T);
  Block blk;
  blk.dyn.assign(shmem + 64, 0);
  blk.waves.resize((T + 63) / 64);
  cur_block() = &blk;
  s.body = [&] { kernel(args...); };
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        for (int t = 0; t < T; ++t) {
          fiber_prepare(s.fib[t]);
          s.fib[t].tl = TL{dim3(t, 0, 0), dim3(bx, by, bz), block, grid};
        }
        run_block(T);
      }
  s.body = nullptr;
  cur_block() = nullptr;
}
}  // namespace emu

#define threadIdx (emu::tl().tid)
#define blockIdx (emu::tl().bid)
#define blockDim (emu::tl().bdim)
#define gridDim (emu::tl().gdim)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  emu::launch((kernel), dim3(grid), dim3(block), (size_t)(shmem), __VA_ARGS__)

// NOTE: kernels must reach every __syncthreads / wave builtin with all threads
// of the block / wave (no early return before one) -- true of good GPU code too.
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
inline void __threadfence_system() {}
inline void __threadfence() {}
inline void __syncthreads() { emu::fiber_yield(emu::FIBER_AT_BLOCK_BARRIER); }

inline unsigned long long __ballot(int pred) {
  emu::Wave& w = emu::my_wave();
  const int lane = threadIdx.x & 63;
  w.slot[lane] = pred ? 1ull : 0ull;
  emu::fiber_yield(emu::FIBER_AT_WAVE_BARRIER);
  const int base = (threadIdx.x / 64) * 64;
  const int n = std::min(64, (int)blockDim.x - base);
  unsigned long long m = 0;
  for (int i = 0; i < n; ++i) m |= (w.slot[i] & 1ull) << i;
  emu::fiber_yield(emu::FIBER_AT_WAVE_BARRIER);
  return m;
}
template <class T>
inline T emu_exchange(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "");
  emu::Wave& w = emu::my_wave();
  const int lane = threadIdx.x & 63;
  unsigned long long bits = 0;
  std::memcpy(&bits, &v, sizeof(T));
  w.slot[lane] = bits;
  emu::fiber_yield(emu::FIBER_AT_WAVE_BARRIER);
  const int base = (threadIdx.x / 64) * 64;
  const int n = std::min(64, (int)blockDim.x - base);
  T out = v;
  if (src_lane >= 0 && src_lane < n) std::memcpy(&out, &w.slot[src_lane], sizeof(T));
  emu::fiber_yield(emu::FIBER_AT_WAVE_BARRIER);
  return out;
}
template <class T> inline T __shfl_down(T v, unsigned d, int width = 64) {
  const int lane = threadIdx.x & 63;
  const int src = lane + (int)d;
  return emu_exchange(v, (src / width == lane / width) ? src : lane);
}
template <class T> inline T __shfl_xor(T v, int m, int width = 64) {
  const int lane = threadIdx.x & 63;
  (void)width;
  return emu_exchange(v, lane ^ m);
}
template <class T> inline T __shfl(T v, int src, int width = 64) {
  const int lane = threadIdx.x & 63;
  return emu_exchange(v, (lane / width) * width + (src % width));
}
inline int __builtin_amdgcn_readlane(int v, int src) { return emu_exchange(v, src); }
inline int __builtin_amdgcn_readfirstlane(int v) { return emu_exchange(v, 0); }  // (callers keep every lane alive)
// v_mov_b32_dpp for the controls the kernels use: row_shr:n (0x110 + n), row_bcast:15 (0x142), row_bcast:31 (0x143).
// A lane whose row is enabled by row_mask and whose source lane exists takes the source's value; every other lane
// keeps `old` (bound_ctrl = false).  bank_mask must be 0xf.
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
  const int lane = threadIdx.x & 63, row = lane >> 4;
  int from = -1;
  if (ctrl > 0x110 && ctrl <= 0x11f) {
    const int n = ctrl - 0x110;
    if ((lane & 15) >= n) from = lane - n;
  } else if (ctrl == 0x142) {
    if (row >= 1) from = row * 16 - 1;
  } else if (ctrl == 0x143) {
    if (row >= 2) from = 31;
  } else {
    std::fprintf(stderr, "emu: DPP control 0x%x not implemented\n", ctrl);
    std::abort();
  }
  if (bank_mask != 0xf) std::abort();
  const int got = emu_exchange(src, from >= 0 ? from : lane);
  const bool enabled = (row_mask >> row) & 1;
  if (enabled && from >= 0) return got;
  return (enabled && bound_ctrl) ? 0 : old;
}
// v_mfma_f64_16x16x4_f64: D (16 x 16) = A (16 x 4) B (4 x 16) + C.  Lane l = (lk = l / 16, li = l % 16) holds A[li][lk],
// B[lk][li] and the four elements D[lk + 4 r][li], r = 0..3 (the layout the product kernels are written for).
#define ext_vector_type(n) vector_size(8 * (n))
typedef double emu_d4 __attribute__((vector_size(32)));
inline emu_d4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, emu_d4 c, int, int, int) {
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  emu_d4 d = c;
  for (int k = 0; k < 4; ++k) {
    const double bk = emu_exchange(b, k * 16 + li);
    for (int r = 0; r < 4; ++r) d[r] += emu_exchange(a, k * 16 + lk + 4 * r) * bk;
  }
  return d;
}
inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __ffs(int x) { return __builtin_ffs(x); }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline double __longlong_as_double(long long x) { double d; std::memcpy(&d, &x, 8); return d; }

template <class T> inline T atomicAdd(T* p, T v) {
  std::lock_guard<std::mutex> g(emu::atomic_mu());
  T old = *p; *p = old + v; return old;
}
template <class T> inline T atomicOr(T* p, T v) {
  std::lock_guard<std::mutex> g(emu::atomic_mu());
  T old = *p; *p = old | v; return old;
}
template <class T> inline T atomicExch(T* p, T v) {
  std::lock_guard<std::mutex> g(emu::atomic_mu());
  T old = *p; *p = v; return old;
}
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_load(p, order, scope) (*(p))
inline void __builtin_amdgcn_s_waitcnt(int) {}
inline void __builtin_amdgcn_sched_barrier(int) {}
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline void __builtin_amdgcn_wave_barrier() { emu::fiber_yield(emu::FIBER_AT_WAVE_BARRIER); }
template <class T> inline T atomicMax(T* p, T v) {
  std::lock_guard<std::mutex> g(emu::atomic_mu());
  T old = *p; if (v > old) *p = v; return old;
}

// ---- host API subset -------------------------------------------------------
inline const char* hipGetErrorString(hipError_t e) { return e == 0 ? "hipSuccess" : "emu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int* v, int attr, int) {
  *v = (attr == hipDeviceAttributeMultiprocessorCount) ? 4 : 160 * 1024;
  return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipMalloc((void**)p, n); }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = nullptr; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t = nullptr) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
template <class F> inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
