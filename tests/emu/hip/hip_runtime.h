// Host-side kernel-LOGIC emulator for the `-m "not gpu"` unit tests.  TEST TOOL ONLY.
//
// It lets the unmodified HIP kernel sources under qiskit-addon-sqd_amd/csrc be
// compiled with g++ (-fsanitize=address) and executed by OS threads, so index /
// sign / bounds bugs are caught in the GPU-less authoring container before a
// gpurun call is spent.  It is NOT a backend: the product loader
// (qiskit_addon_sqd_amd/_capi.py) only ever loads the hipcc-built
// libsqd_hip.so, and tests/emu builds into tests/emu/_build/libsqd_emu.so which
// nothing outside tests/ knows about.
//
// Model: one OS thread per GPU thread of a block; the blocks of a grid run one
// after another; wave = 64 consecutive threads; wave-level builtins rendezvous
// on a per-wave barrier, __syncthreads on a per-block barrier.  "Device memory"
// is host memory.  Streams and events are no-ops.
#pragma once
#include <algorithm>
#include <atomic>
#include <barrier>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct uint2 { unsigned x, y; };
struct double2 { double x, y; };
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }

typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum { hipDeviceAttributeMultiprocessorCount = 1, hipDeviceAttributeMaxSharedMemoryPerBlock = 2 };
enum { hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define HIP_DYNAMIC_SHARED(type, var) type* var = reinterpret_cast<type*>(emu::dyn_smem());

namespace emu {
struct Wave {
  std::unique_ptr<std::barrier<>> bar;
  unsigned long long slot[64];
};
struct Block {
  std::unique_ptr<std::barrier<>> bar;
  std::vector<Wave> waves;
  std::vector<char> dyn;
};
inline Block*& cur_block() { static Block* b = nullptr; return b; }
struct TL { dim3 tid, bid, bdim, gdim; };
inline TL& tl() { static thread_local TL t; return t; }
inline void* dyn_smem() {
  // 16-byte aligned start
  auto p = reinterpret_cast<uintptr_t>(cur_block()->dyn.data());
  return reinterpret_cast<void*>((p + 15) & ~uintptr_t(15));
}
inline Wave& my_wave() { return cur_block()->waves[tl().tid.x / 64]; }
inline std::mutex& atomic_mu() { static std::mutex m; return m; }
inline std::mutex& launch_mu() { static std::mutex m; return m; }

// Worker threads are kept between launches (creating and joining up to 1024 OS threads per launch was most of the
// emulator's run time: a Davidson iteration is four launches).  Never joined: the pool lives as long as the process.
struct Pool {
  std::mutex mu;
  std::condition_variable cv_start, cv_done;
  std::vector<std::thread> threads;
  std::function<void(int)> job;
  long gen = 0;
  int active = 0, done = 0;
  void worker(int id) {
    long seen = 0;
    for (;;) {
      std::function<void(int)> j;
      bool mine;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_start.wait(lk, [&] { return gen != seen; });
        seen = gen;
        mine = id < active;
        if (mine) j = job;
      }
      if (!mine) continue;
      j(id);
      std::lock_guard<std::mutex> lk(mu);
      if (++done == active) cv_done.notify_one();
    }
  }
  void run(int T, std::function<void(int)> f) {
    {
      std::lock_guard<std::mutex> lk(mu);
      while ((int)threads.size() < T) {
        const int id = (int)threads.size();
        threads.emplace_back([this, id] { worker(id); });
        threads.back().detach();
      }
      job = std::move(f);
      active = T;
      done = 0;
      ++gen;
    }
    cv_start.notify_all();
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return done == active; });
  }
};
inline Pool& pool() { static Pool* p = new Pool(); return *p; }

template <class K, class... A>
void launch(K kernel, dim3 grid, dim3 block, size_t shmem, A... args) {
  // one kernel at a time: the current block and the `__shared__ static` variables are process-wide, and host
  // threads (solve_sci_batch runs several contexts concurrently) may launch at the same moment
  std::lock_guard<std::mutex> launch_guard(launch_mu());
  const int T = block.x;
  Block blk;
  blk.dyn.assign(shmem + 64, 0);
  const int nw = (T + 63) / 64;
  blk.waves.resize(nw);
  blk.bar.reset(new std::barrier<>(T));
  for (int w = 0; w < nw; ++w) blk.waves[w].bar.reset(new std::barrier<>(std::min(64, T - 64 * w)));
  cur_block() = &blk;
  std::barrier<> block_seq(T);  // all threads move from block b to block b+1 together
  pool().run(T, [&](int t) {
    TL& x = tl();
    x.bdim = block;
    x.gdim = grid;
    x.tid = dim3(t, 0, 0);
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        x.bid = dim3(bx, by, 0);
        kernel(args...);
        block_seq.arrive_and_wait();
      }
  });
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
inline void __syncthreads() { emu::cur_block()->bar->arrive_and_wait(); }

inline unsigned long long __ballot(int pred) {
  emu::Wave& w = emu::my_wave();
  const int lane = threadIdx.x & 63;
  w.slot[lane] = pred ? 1ull : 0ull;
  w.bar->arrive_and_wait();
  const int base = (threadIdx.x / 64) * 64;
  const int n = std::min(64, (int)blockDim.x - base);
  unsigned long long m = 0;
  for (int i = 0; i < n; ++i) m |= (w.slot[i] & 1ull) << i;
  w.bar->arrive_and_wait();
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
  w.bar->arrive_and_wait();
  const int base = (threadIdx.x / 64) * 64;
  const int n = std::min(64, (int)blockDim.x - base);
  T out = v;
  if (src_lane >= 0 && src_lane < n) std::memcpy(&out, &w.slot[src_lane], sizeof(T));
  w.bar->arrive_and_wait();
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
inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(long long x) { return __builtin_ffsll(x); }
inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }
inline double __longlong_as_double(long long x) { double d; std::memcpy(&d, &x, 8); return d; }

template <class T> inline T atomicAdd(T* p, T v) {
  std::lock_guard<std::mutex> g(emu::atomic_mu());
  T old = *p; *p = old + v; return old;
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
#define __builtin_amdgcn_fence(order, scope) ((void)0)
inline void __builtin_amdgcn_wave_barrier() { emu::my_wave().bar->arrive_and_wait(); }
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
