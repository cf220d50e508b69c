// (not compiled) sqd_spmm.hip @0358935: the row-grouped product with its records through LDS
// ---- the same product with the group's records through LDS instead of the scalar cache.  The four wavefronts of a
// workgroup take FOUR ADJACENT PANELS of one group: its list {source, 8 coefficients} is fetched once per workgroup with
// coalesced vector loads, SC sources at a time into a double-buffered LDS slab, and read back with uniform-address
// (broadcast) LDS reads -- k_spmm_grouped streams 68 bytes per source and wavefront through the scalar cache, whose miss
// path (counters: waves parked 72 % of their life) is what bounds it.
constexpr int SC = 32;  // sources per LDS slab (GPAD is a multiple)
static_assert(GPAD % SC == 0 || SC % GPAD == 0, "slabs and the lists' padding must nest");
template <int GJ>
__global__ void __launch_bounds__(256) k_spmm_grouped_lds(const GroupedArgs g) {
  __shared__ double s_coef[2][SC * GR];
  __shared__ uint32_t s_src[2][SC];
  if (g.stop && *g.stop) return;
  const int side = blockIdx.y;
  const unsigned ng = g.ngroups[side], np = g.npanels[side], nq = (np + 3u) / 4u;  // panel quads
  const int64_t n = g.n[side], m = g.m[side];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  unsigned quad, r;
  if (g.xcd_split) {
    const unsigned x = blockIdx.x & 7u, q = blockIdx.x >> 3;
    quad = (q / ng) * 8u + x;
    r = q % ng;
  } else {
    quad = blockIdx.x / ng;
    r = blockIdx.x % ng;
  }
  if (quad >= nq) return;  // (uniform over the workgroup)
  const int64_t grp = (int64_t)g.order[side][r];
  const unsigned panel = quad * 4u + (unsigned)__builtin_amdgcn_readfirstlane(wave);
  const bool wave_on = panel < np;  // (a wavefront past the last panel still helps staging and meets the barriers)
  const double* __restrict__ in = g.in[side];
  if (side == 0 && g.vec_index) in += (int64_t)(*g.vec_index - 1) * g.in_stride;
  const int64_t base = g.base[side][grp];
  const uint32_t cnt = g.cnt[side][grp];  // (a multiple of GPAD)
  const uint32_t* __restrict__ src = g.src[side] + base;
  const double* __restrict__ coef = g.coef[side] + base * GR;
  const unsigned c0 = (wave_on ? panel : 0u) * (unsigned)(64 * GJ);
  bool ok[GJ];
  unsigned col[GJ];
#pragma unroll
  for (int j = 0; j < GJ; ++j) {
    ok[j] = wave_on && (int64_t)c0 + j * 64 + lane < m;
    col[j] = ((int64_t)c0 + j * 64 + lane < m) ? (unsigned)(j * 64 + lane) : (unsigned)(m - 1 - c0);
  }
  in += c0;
  double acc[GR][GJ];
#pragma unroll
  for (int i = 0; i < GR; ++i)
#pragma unroll
    for (int j = 0; j < GJ; ++j) acc[i][j] = 0.0;
  // slab loads: thread t holds coefficient t of the slab (SC * GR = 256 of them) and, t < SC, source t
  double pc = coef[tid];
  uint32_t ps = src[tid < SC ? tid : 0];
  const uint32_t nslab = cnt / SC + ((cnt % SC) ? 1u : 0u);
  for (uint32_t sl = 0; sl < nslab; ++sl) {
    const int b = (int)(sl & 1u);
    s_coef[b][tid] = pc;
    if (tid < SC) s_src[b][tid] = ps;
    __syncthreads();
    if (sl + 1 < nslab) {  // (the list ends with GPAD records of padding: a whole slab may be read past cnt)
      pc = coef[(int64_t)(sl + 1) * SC * GR + tid];
      ps = src[(int64_t)(sl + 1) * SC + (tid < SC ? tid : 0)];
    }
    if (wave_on) {
      const uint32_t left = cnt - sl * SC, ns = left < (uint32_t)SC ? left : (uint32_t)SC;  // (a multiple of 16)
      for (uint32_t u0 = 0; u0 < ns; u0 += 16) {
        double x[16][GJ];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const uint32_t sv = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_src[b][u0 + u]);
          const double* __restrict__ rowp = in + (int64_t)sv * m;
#pragma unroll
          for (int j = 0; j < GJ; ++j) x[u][j] = spmm_ldu(rowp, col[j]);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const double* cp = &s_coef[b][(u0 + u) * GR];
          double cf[GR];
#pragma unroll
          for (int i = 0; i < GR; i += 2) {
            const double2 c2 = *reinterpret_cast<const double2*>(cp + i);
            cf[i] = c2.x;
            cf[i + 1] = c2.y;
          }
#pragma unroll
          for (int i = 0; i < GR; ++i)
#pragma unroll
            for (int j = 0; j < GJ; ++j) acc[i][j] += cf[i] * x[u][j];
        }
      }
    }
  }
  double* __restrict__ out = g.out[side] + c0;
#pragma unroll
  for (int j = 0; j < GJ; ++j)
    if (ok[j])
#pragma unroll
      for (int i = 0; i < GR; ++i)
        if (grp * GR + i < n) out[(grp * GR + i) * m + col[j]] = acc[i][j];
}


// ---- host side (spmm_launch excerpt)
#if 0
    // SQD_SPMM_LDS=1: the group records through an LDS slab shared by four panels (k_spmm_grouped_lds) instead of the
    // scalar cache.  Measured SLOWER (profiles/r05/variants_probe.txt: the product alone 117 vs 97 us at 1000^2, 1017 vs
    // 855 at 3000^2): three uniform-address LDS reads and a barrier per slab cost more than the scalar stream they replace.
    static const bool lds_records = [] {
      const char* env = std::getenv("SQD_SPMM_LDS");
      return env && std::atoi(env) != 0;
    }();
    if (lds_records && gj <= 2) {
      unsigned gxl = 1;
      for (int sp = 0; sp < 2; ++sp) {
        const unsigned nq = (gg.npanels[sp] + 3u) / 4u;
        const uint64_t blocks = gg.xcd_split ? 8ull * ((nq + 7u) / 8u) * gg.ngroups[sp] : (uint64_t)nq * gg.ngroups[sp];
        gxl = blocks > gxl ? (unsigned)blocks : gxl;
      }
      if (gj == 2) hipLaunchKernelGGL((k_spmm_grouped_lds<2>), dim3(gxl, 2), dim3(256), 0, c->stream, gg);
      else hipLaunchKernelGGL((k_spmm_grouped_lds<1>), dim3(gxl, 2), dim3(256), 0, c->stream, gg);

#endif
