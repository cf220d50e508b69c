// Small device-side helpers shared by the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace sqd {

// ---- wave64 reductions on the DPP path.  `__shfl_down` of a double is two ds_bpermute_b32 -- an LDS-crossbar round
// trip of ~120 clocks per step, seven dependent steps per reduction: the phase clocks of round 3 (profiles/probes/
// _phase_clock.py) put 3-4 us of a 9 us BLAS-1 kernel into its block reduction and a third of the projected eigenproblem
// into its dot products.  v_mov_b32_dpp moves a lane's register to a neighbour inside the VALU (no LDS): row_shr 1, 2, 4,
// 8 leave the sum of a row of 16 lanes in its lane 15, row_bcast:15 (rows 1 and 3) and row_bcast:31 (rows 2 and 3) carry
// the row totals up, and lane 63 holds the sum of the wavefront.  Lanes without a source take `fill`.  One fixed tree
// => bitwise reproducible.  All 64 lanes must be active.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_take(double v, double fill) {
  int w[2], f[2];
  __builtin_memcpy(w, &v, 8);
  __builtin_memcpy(f, &fill, 8);
  w[0] = __builtin_amdgcn_update_dpp(f[0], w[0], CTRL, ROW_MASK, 0xf, false);
  w[1] = __builtin_amdgcn_update_dpp(f[1], w[1], CTRL, ROW_MASK, 0xf, false);
  double r;
  __builtin_memcpy(&r, w, 8);
  return r;
}
__device__ inline double wave_sum_lane63(double v) {  // the total on lane 63 (other lanes: partial sums)
  v += dpp_take<0x111, 0xf>(v, 0.0);
  v += dpp_take<0x112, 0xf>(v, 0.0);
  v += dpp_take<0x114, 0xf>(v, 0.0);
  v += dpp_take<0x118, 0xf>(v, 0.0);
  v += dpp_take<0x142, 0xa>(v, 0.0);
  v += dpp_take<0x143, 0xc>(v, 0.0);
  return v;
}
__device__ inline double wave_max_lane63(double v) {
  double o;
  o = dpp_take<0x111, 0xf>(v, v), v = o > v ? o : v;
  o = dpp_take<0x112, 0xf>(v, v), v = o > v ? o : v;
  o = dpp_take<0x114, 0xf>(v, v), v = o > v ? o : v;
  o = dpp_take<0x118, 0xf>(v, v), v = o > v ? o : v;
  o = dpp_take<0x142, 0xa>(v, v), v = o > v ? o : v;
  o = dpp_take<0x143, 0xc>(v, v), v = o > v ? o : v;
  return v;
}

// Sum over the workgroup; result valid on thread 0.  `red` = >= 16 doubles of LDS.
// Fixed tree + fixed wave order => bitwise reproducible.
__device__ inline double block_sum(double v, double* red) {
  v = wave_sum_lane63(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 63) red[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 0; w < nw; ++w) s += red[w];
  }
  return s;
}

// Sum N per-thread values over the workgroup with ONE barrier: DPP tree inside each wave (all N values, branch-free:
// N independent chains that the scheduler interleaves -- a branch per value made them N chains in sequence), lane 63
// parks the wave totals in LDS (red[wave*N + v]), thread v adds the waves in order: `block_sum_multi_get`.
// Values at or past n must be zero (or are never read).
template <int N>
__device__ inline void block_sum_multi(double (&vals)[N], int n, double* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  (void)n;
#pragma unroll
  for (int v = 0; v < N; ++v) {
    const double x = wave_sum_lane63(vals[v]);
    if (lane == 63) red[wave * N + v] = x;
    if (v % 4 == 3) __builtin_amdgcn_sched_barrier(0);  // four chains interleaved at a time (registers)
  }
  __syncthreads();
}
// after block_sum_multi: total of value v (any thread may call; v < n)
template <int N>
__device__ inline double block_sum_multi_get(const double* red, int v) {
  const int nw = (blockDim.x + 63) >> 6;
  double s = 0.0;
  for (int w = 0; w < nw; ++w) s += red[w * N + v];
  return s;
}

// Device-coherent accesses for data handed from one workgroup to another INSIDE a kernel.  On gfx950 the
// 8 XCDs have private L2s: ordinary stores may sit dirty in the writer's L2, and an agent-scope fence
// (__threadfence) writes the whole L2 back -- measured at ~0.2 us per workgroup, serialised.  Relaxed
// agent-scope atomics go through to the coherence point (sc1) for just the words concerned.
__device__ inline void coherent_store(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double coherent_load(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// "The workgroup that arrives last finishes the job": two-level arrival count (COUNT_GROUPS group words + a top
// word, each in its own 128-byte line -- device-scope atomics on ONE word serialise at ~50 ns each, and atomics on
// one line serialise in one L2 channel).  Call with every thread of the workgroup AFTER its hand-over data has
// been written with coherent_store (the waitcnt below completes those stores; no L2-wide fence); returns true in
// every thread of the one workgroup that arrived last, which then reads the others' data with coherent_load.
// The counters reset themselves, so one zeroed block of COUNT_WORDS words serves every fused reduction on a stream.
constexpr unsigned COUNT_GROUPS = 16;
constexpr unsigned COUNT_STRIDE = 32;
constexpr int COUNT_WORDS = (int)((COUNT_GROUPS + 1) * COUNT_STRIDE);
__device__ inline bool arrive_last(unsigned* counter, unsigned block, unsigned nblocks) {
  __shared__ int s_last;
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    int last = 0;
    const unsigned G = COUNT_GROUPS, grp = block % G;
    const unsigned gsize = (nblocks - grp + G - 1) / G, ngroups = nblocks < G ? nblocks : G;
    if (atomicAdd(&counter[COUNT_STRIDE * (1 + grp)], 1u) == gsize - 1) {
      atomicExch(&counter[COUNT_STRIDE * (1 + grp)], 0u);
      if (atomicAdd(&counter[0], 1u) == ngroups - 1) {
        atomicExch(&counter[0], 0u);
        last = 1;
      }
    }
    s_last = last;
  }
  __syncthreads();
  return s_last != 0;
}

__device__ inline void coherent_store_i64(int64_t* p, int64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline int64_t coherent_load_i64(const int64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Payload words of the host-visible mailbox (fine-grained pinned host memory): written through at system
// scope.  Protocol of a post: every writing thread issues its stores and waits for them (s_waitcnt), the
// workgroup meets at a barrier, then ONE thread fences at system scope and writes the sequence word.  (Fencing
// in every thread -- eight waves each writing the L2 back -- is what this replaces.)
__device__ inline void mail_store(double* p, double v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace sqd

namespace sqd {
// ---- one-wavefront collectives (the small dense solves that run inside the LAST workgroup of a fused
// reduction kernel).  All 64 lanes must call them.  wave_sync: LDS written by one lane is read by another.
__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
// value of lane `src` (uniform over the wave) on every lane: v_readlane_b32 x2, no LDS crossbar
__device__ inline double wave_bcast(double v, int src);
__device__ inline double wave_sum(double v) {  // fixed tree => bitwise reproducible; result on every lane
  return wave_bcast(wave_sum_lane63(v), 63);
}
__device__ inline double wave_max(double v) { return wave_bcast(wave_max_lane63(v), 63); }
__device__ inline double wave_bcast(double v, int src) {
  int w[2];
  __builtin_memcpy(w, &v, 8);
  w[0] = __builtin_amdgcn_readlane(w[0], src);
  w[1] = __builtin_amdgcn_readlane(w[1], src);
  double r;
  __builtin_memcpy(&r, w, 8);
  return r;
}
// 1/x to ~1 ulp from the hardware reciprocal + one Newton step: 4 instructions instead of the ~35 of an IEEE
// division.  For the shifted solves of the Rayleigh-quotient iteration only (self-correcting: the acceptance test
// looks at the residual of the final pair, which is computed with ordinary arithmetic).
__device__ inline double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  return r * (2.0 - x * r);
}
// index of the largest value (ties to the lower index); lanes without a candidate pass a negative value
__device__ inline int wave_argmax(double v, int idx) {
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_down(v, off);
    const int oi = __shfl_down(idx, off);
    if (ov > v || (ov == v && oi < idx)) {
      v = ov;
      idx = oi;
    }
  }
  return __shfl(idx, 0);
}

// (value, index) minimum over the workgroup, ties to the lower index; result valid on thread 0.
// Lanes without a candidate pass index -1.
__device__ inline void block_argmin(double& best, int64_t& bi) {
  __shared__ double s_v[16];
  __shared__ long long s_i[16];
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_down(best, off);
    const long long oi = __shfl_down((long long)bi, off);
    if (oi >= 0 && (bi < 0 || ov < best || (ov == best && oi < bi))) {
      best = ov;
      bi = oi;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) {
    s_v[wave] = best;
    s_i[wave] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int w = 1; w < nw; ++w)
      if (s_i[w] >= 0 && (bi < 0 || s_v[w] < best || (s_v[w] == best && s_i[w] < bi))) {
        best = s_v[w];
        bi = s_i[w];
      }
  }
}
}  // namespace sqd
