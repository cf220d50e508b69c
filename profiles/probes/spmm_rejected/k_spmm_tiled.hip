// (not compiled) sqd_spmm.hip @0358935: the tiled same-spin product
// tiled form of the same lists (k_spmm_tiled): row t's links grouped by source chunk of TS rows, every group padded to a
// multiple of 4 records {byte offset of the source row inside the staged tile, value}; beg / cnt [chunk][row]
struct SpmmTiles {
  DevBuf beg, cnt, off, val, rowbase;
  std::vector<int64_t> h_rowbase;
  int nch = 0;
  int64_t npad = 0, cap = 0;
};

// ---- the product through LDS tiles (the default): the same sum, with the operand rows of a 64-column panel staged in LDS
// TS = 128 source rows at a time and shared by the TB = 128 target rows of the workgroup (8 wavefronts x 16 targets, one
// accumulator per target and lane), so that a staged element serves TB x density multiply-adds (7 at 3000 strings,
// 14 at 1000) instead of one, and the vector L1 -- the bound of k_spmm_rows above -- carries 1/7 .. 1/14 of the bytes.
// A link costs one scalar record (4-byte tile offset + 8-byte value, through the scalar cache, four records per
// request), one conflict-free ds_read_b64 (64 lanes = 64 consecutive columns of one staged row) and one multiply-add.
// The next tile's 16 elements per thread are requested before the current tile is consumed (register-staged double
// buffering).  Fixed order (chunks ascending; inside a chunk singles, then doubles, by source): the same bits on every run.
constexpr int TS = 128, TB = 128, TW = 8, TPANEL = 64, TTHREADS = 1024, TGRAN = 4;
struct TileBuildArgs {
  int64_t n[2];
  int nch[2];
  int64_t npad[2];
  GPtr<const int64_t> s_ptr[2], d_ptr[2];
  GPtr<const SRec> s_rec[2];
  GPtr<const double> s_val[2], d_val[2];
  GPtr<const uint32_t> d_src[2];
  GPtr<const int64_t> rowbase[2];
  GPtr<uint32_t> beg[2], cnt[2], off[2];
  GPtr<double> val[2];
};
// one wavefront per row: histogram of the row's links over the source chunks (LDS), padded group sizes, their prefix
// sums, then every link to its place: group start + rank inside the group (the CSR lists are sorted by source, so the
// rank of a link is its index minus the index of the group's first link)
__global__ void __launch_bounds__(256) k_spmm_tile_build(const TileBuildArgs g) {
  __shared__ int hist[4][2][256];   // [wave][singles | doubles][chunk]: counts, then exclusive prefix (first link of the chunk)
  __shared__ int gstart[4][256];    // group start (records, relative to the row's region)
  __shared__ int gtot[4][256];      // unpadded group size
  const int sd = blockIdx.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n = g.n[sd];
  const int nch = g.nch[sd];
  const int64_t t = (int64_t)blockIdx.x * 4 + w;
  const bool live = t < n;
  for (int c = lane; c < 256; c += 64) hist[w][0][c] = hist[w][1][c] = 0;
  __syncthreads();
  const int64_t* __restrict__ sp = g.s_ptr[sd];
  const int64_t* __restrict__ dp = g.d_ptr[sd];
  const int64_t s0 = live ? sp[t] : 0, ns = live ? sp[t + 1] - s0 : 0, d0 = live ? dp[t] : 0, nd = live ? dp[t + 1] - d0 : 0;
  const SRec* __restrict__ rec = g.s_rec[sd];
  const uint32_t* __restrict__ ds = g.d_src[sd];
  for (int64_t k = lane; k < ns; k += 64) atomicAdd(&hist[w][0][rec[s0 + k].src / TS], 1);
  for (int64_t k = lane; k < nd; k += 64) atomicAdd(&hist[w][1][ds[d0 + k] / TS], 1);
  __syncthreads();
  // lane l owns chunks 4 l .. 4 l + 3: exclusive prefixes of the singles' counts, the doubles' counts, the padded sizes
  int cs[4], cd[4], cp[4], as = 0, ad = 0, ap = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = 4 * lane + u;
    cs[u] = hist[w][0][c];
    cd[u] = hist[w][1][c];
    cp[u] = (cs[u] + cd[u] + TGRAN - 1) / TGRAN * TGRAN;
    as += cs[u];
    ad += cd[u];
    ap += cp[u];
  }
  int xs = as, xd = ad, xp = ap;  // inclusive scans over the lanes
  for (int d = 1; d < 64; d <<= 1) {
    const int ys = __shfl(xs, lane - d), yd = __shfl(xd, lane - d), yp = __shfl(xp, lane - d);
    if (lane >= d) {
      xs += ys;
      xd += yd;
      xp += yp;
    }
  }
  xs -= as;
  xd -= ad;
  xp -= ap;
  __syncthreads();
  const int64_t base = live ? g.rowbase[sd][t] : 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = 4 * lane + u;
    hist[w][0][c] = xs;
    hist[w][1][c] = xd;
    gstart[w][c] = xp;
    gtot[w][c] = cs[u] + cd[u];
    if (live && c < nch) {
      g.beg[sd][(int64_t)c * g.npad[sd] + t] = (uint32_t)(base + xp);
      g.cnt[sd][(int64_t)c * g.npad[sd] + t] = (uint32_t)cp[u];
    }
    xs += cs[u];
    xd += cd[u];
    xp += cp[u];
  }
  __syncthreads();
  if (!live) return;
  uint32_t* __restrict__ off = g.off[sd] + base;
  double* __restrict__ val = g.val[sd] + base;
  const double* __restrict__ sv = g.s_val[sd];
  const double* __restrict__ dv = g.d_val[sd];
  for (int64_t k = lane; k < ns; k += 64) {
    const uint32_t src = rec[s0 + k].src;
    const int c = (int)(src / TS);
    const int pos = gstart[w][c] + ((int)k - hist[w][0][c]);
    off[pos] = (src % TS) * (uint32_t)(TPANEL * 8);
    val[pos] = sv[s0 + k];
  }
  for (int64_t k = lane; k < nd; k += 64) {
    const uint32_t src = ds[d0 + k];
    const int c = (int)(src / TS);
    const int nsc = (c + 1 < 256 ? hist[w][0][c + 1] : (int)ns) - hist[w][0][c];  // singles of this group
    const int pos = gstart[w][c] + nsc + ((int)k - hist[w][1][c]);
    off[pos] = (src % TS) * (uint32_t)(TPANEL * 8);
    val[pos] = dv[d0 + k];
  }
  for (int c = lane; c < nch; c += 64) {
    const int tot = gtot[w][c], pad = (tot + TGRAN - 1) / TGRAN * TGRAN;
    for (int p = tot; p < pad; ++p) {
      off[gstart[w][c] + p] = 0u;
      val[gstart[w][c] + p] = 0.0;
    }
  }
}

struct TiledArgs {
  GPtr<const uint32_t> beg[2], cnt[2], off[2];
  GPtr<const double> val[2];
  GPtr<const double> in[2];
  GPtr<double> out[2];
  int64_t n[2], m[2], npad[2];
  int nch[2];
  unsigned ntb[2], npanels[2];
  GPtr<const int> stop, vec_index;
  int64_t in_stride;
};
// Where the link records come from decides this kernel.  First version: scalar loads (s_load_dwordx4 / x8 per four links)
// straight from the row-major lists -- every (target, chunk) group then starts with a scalar-cache miss that nothing
// hides (groups hold 7 links on average): 2.24 ms at 3000 x 3000 against 1.35 ms for k_spmm_rows.  Now the records of a
// wavefront's targets travel like the tile: one coalesced vector load per target (lane l = record l of the group, up to
// TSL = 32 per pass), requested a chunk ahead, parked in the wavefront's own LDS slab, and read back with uniform-address
// (broadcast) LDS reads -- four offsets per ds_read_b128, four values per two.  Per link: 3 LDS cycles of records, 2 of
// operand, one address add, one multiply-add.
constexpr int TSL = 32;  // records per target and pass in the LDS slab
static_assert(TGRAN == 4, "the record reads of k_spmm_tiled are written for groups of four");
__global__ void __launch_bounds__(TTHREADS) k_spmm_tiled(const TiledArgs g) {
  HIP_DYNAMIC_SHARED(double, smem)  // [TS][TPANEL] tile | per wavefront: values [TW][TSL] | offsets [TW][TSL]
  if (g.stop && *g.stop) return;
  const int side = blockIdx.y;
  // workgroup b runs on XCD b mod 8: the target blocks of one column panel share an XCD (its L2 holds the panel)
  const unsigned x = blockIdx.x & 7u, q = blockIdx.x >> 3, ntb = g.ntb[side];
  const unsigned panel = (q / ntb) * 8u + x, tb = q % ntb;
  if (panel >= g.npanels[side]) return;
  const int64_t n = g.n[side], m = g.m[side], npad = g.npad[side];
  const int nch = g.nch[side];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t t0 = (int64_t)tb * TB + wave * TW;
  const double* __restrict__ in = g.in[side];
  if (side == 0 && g.vec_index) in += (int64_t)(*g.vec_index - 1) * g.in_stride;
  const bool wave_on = t0 < n;  // (uniform; a wavefront past the last row still stages the tile and meets the barriers)
  const uint32_t* __restrict__ beg = g.beg[side] + (wave_on ? t0 : 0);
  const uint32_t* __restrict__ cnt = g.cnt[side] + (wave_on ? t0 : 0);
  const uint32_t* __restrict__ off = g.off[side];
  const double* __restrict__ val = g.val[side];
  const int64_t col = (int64_t)panel * TPANEL + lane;
  const bool col_ok = col < m;
  const int64_t colc = col_ok ? col : m - 1;
  double* tile = smem;
  double* rval = smem + TS * TPANEL + wave * (TW * TSL);
  uint32_t* roff = reinterpret_cast<uint32_t*>(smem + TS * TPANEL + (TTHREADS / 64) * (TW * TSL)) + wave * (TW * TSL);
  constexpr int NL = TS * TPANEL / TTHREADS;  // tile elements per thread: rows (tid >> 6) + (TTHREADS / 64) i
  double pf[NL];
  auto fetch_tile = [&](int c) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int64_t r = (int64_t)c * TS + (tid >> 6) + (TTHREADS / 64) * i;
      const double v = in[(r < n ? r : n - 1) * m + colc];
      pf[i] = (r < n && col_ok) ? v : 0.0;
    }
  };
  // link groups of this wavefront's TW targets in one chunk: first record and padded count (scalars); pass p of the
  // records: lane l < TSL holds record TSL p + l of every target's group (zero weight past the group's end)
  uint32_t bcur[TW], ncur[TW], bnxt[TW], nnxt[TW];
  uint32_t po[TW];
  double pv[TW];
  auto fetch_heads = [&](int c, uint32_t* b, uint32_t* nn) {
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      b[i] = beg[(int64_t)c * npad + i];
      const uint32_t cn = cnt[(int64_t)c * npad + i];
      nn[i] = wave_on ? cn : 0u;
    }
  };
  auto fetch_recs = [&](const uint32_t* b, const uint32_t* nn, uint32_t p) {
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      const uint32_t j = p * TSL + (uint32_t)(lane & (TSL - 1));
      const bool on = j < nn[i];
      const uint32_t a = b[i] + (on ? j : 0u);
      const uint32_t ov = off[a];
      const double vv = val[a];
      po[i] = on ? ov : 0u;
      pv[i] = on ? vv : 0.0;
    }
  };
  auto park_recs = [&]() {
    if (lane < TSL) {
#pragma unroll
      for (int i = 0; i < TW; ++i) {
        roff[i * TSL + lane] = po[i];
        rval[i * TSL + lane] = pv[i];
      }
    }
  };
  double acc[TW];
#pragma unroll
  for (int i = 0; i < TW; ++i) acc[i] = 0.0;
  const char* tile_lane = reinterpret_cast<const char*>(tile) + lane * 8;
  // records of `cn` (<= TSL, a multiple of TGRAN) links of target i from the slab
  auto consume = [&](int i, uint32_t cn, double a) {
    const uint32_t* ro = roff + i * TSL;
    const double* rv = rval + i * TSL;
    for (uint32_t k = 0; k < cn; k += TGRAN) {
      // (16-byte reads: ds_read_b128 is 4 LDS cycles where the ds_read2_b64 the compiler picks for 8-byte-aligned
      // pointers is 8; the slab and k are multiples of 16 bytes)
      const uint4 o4 = *reinterpret_cast<const uint4*>(ro + k);
      const double2 v01 = *reinterpret_cast<const double2*>(rv + k), v23 = *reinterpret_cast<const double2*>(rv + k + 2);
      const uint32_t o[TGRAN] = {o4.x, o4.y, o4.z, o4.w};
      const double v[TGRAN] = {v01.x, v01.y, v23.x, v23.y};
      double xv[TGRAN];
#pragma unroll
      for (int u = 0; u < TGRAN; ++u) xv[u] = *reinterpret_cast<const double*>(tile_lane + o[u]);
#pragma unroll
      for (int u = 0; u < TGRAN; ++u) a += v[u] * xv[u];
    }
    return a;
  };
  fetch_tile(0);
  fetch_heads(0, bcur, ncur);
  fetch_recs(bcur, ncur, 0);
  if (nch > 1) fetch_heads(1, bnxt, nnxt);
  for (int c = 0; c < nch; ++c) {
#pragma unroll
    for (int i = 0; i < NL; ++i) tile[((tid >> 6) + (TTHREADS / 64) * i) * TPANEL + lane] = pf[i];
    park_recs();
    __syncthreads();
    // requests of the next chunk: its tile rows, the first pass of its records (heads already here), the heads after it
    uint32_t nthis[TW], bthis[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      nthis[i] = ncur[i];
      bthis[i] = bcur[i];
    }
    uint32_t nmax = 0;
#pragma unroll
    for (int i = 0; i < TW; ++i) nmax = nthis[i] > nmax ? nthis[i] : nmax;
    const bool single_pass = nmax <= TSL;  // (uniform; the rule, not the exception: groups hold 4-16 links)
    if (c + 1 < nch) {
      fetch_tile(c + 1);
#pragma unroll
      for (int i = 0; i < TW; ++i) {
        bcur[i] = bnxt[i];
        ncur[i] = nnxt[i];
      }
      if (single_pass) fetch_recs(bcur, ncur, 0);
      if (c + 2 < nch) fetch_heads(c + 2, bnxt, nnxt);
    }
#pragma unroll
    for (int i = 0; i < TW; ++i) acc[i] = consume(i, nthis[i] < (uint32_t)TSL ? nthis[i] : (uint32_t)TSL, acc[i]);
    if (!single_pass) {
      // long groups (rows of the Hartree-Fock neighbourhood): further passes over the same tile, their records loaded
      // on the spot; the slab is private to the wavefront, so only its own LDS accesses need ordering
      for (uint32_t p = 1; p * TSL < nmax; ++p) {
        fetch_recs(bthis, nthis, p);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        park_recs();
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < TW; ++i) {
          const uint32_t done = p * TSL, left = nthis[i] > done ? nthis[i] - done : 0u;
          acc[i] = consume(i, left < (uint32_t)TSL ? left : (uint32_t)TSL, acc[i]);
        }
      }
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      if (c + 1 < nch) fetch_recs(bcur, ncur, 0);  // (po / pv were in use: the next chunk's first pass only now)
    }
    __syncthreads();
  }
  if (col_ok) {
    double* __restrict__ out = g.out[side];
#pragma unroll
    for (int i = 0; i < TW; ++i)
      if (t0 + i < n) out[(t0 + i) * m + col] = acc[i];
  }
}


// ---- host side (spmm_build / spmm_launch excerpts)
#if 0
    s->tiled = false;
    return SQD_OK;
  }
  // the tiled form (SQD_SPMM_GROUPED=0 SQD_SPMM_TILED=1; otherwise k_spmm_rows on the merged lists above).  Chunks of TS source rows: at most
  // 256 per side (the build kernel's histogram), i.e. 32 768 strings per spin.
  static const bool tiled_env = [] {
    const char* env = std::getenv("SQD_SPMM_TILED");
    return env && std::atoi(env) != 0;
  }();
  s->tiled = tiled_env && maxn <= 256 * TS;
  if (!s->tiled) return SQD_OK;
  TileBuildArgs tb;
  for (int sp = 0; sp < 2; ++sp) {
    const SpinTables& t = c->sp[sp];
    SpmmTiles& tl = s->tiles[sp];
    const int64_t* ps = hptr[sp][0];
    const int64_t* pd = hptr[sp][1];
    tl.nch = (int)((t.n + TS - 1) / TS);
    tl.npad = (t.n + TW - 1) / TW * TW + TW;
    // a row's region: its links + the worst-case padding (TGRAN - 1 per non-empty group), cut on the host from the CSR
    // pointers it holds anyway -- no device scan, no second synchronisation
    tl.h_rowbase.resize((size_t)t.n + 1);
    tl.h_rowbase[0] = 0;
    for (int64_t i = 0; i < t.n; ++i) {
      const int64_t len = (ps[i + 1] - ps[i]) + (pd[i + 1] - pd[i]);
      const int64_t groups = len < tl.nch ? len : tl.nch;
      tl.h_rowbase[i + 1] = tl.h_rowbase[i] + (len + (TGRAN - 1) * groups + TGRAN - 1) / TGRAN * TGRAN;
    }
    tl.cap = tl.h_rowbase[t.n];
    if (tl.cap + TGRAN > 0xffffffffll) {
      s->tiled = false;
      return SQD_OK;
    }
    SQD_TRY(tl.rowbase.reserve((size_t)(t.n + 1) * 8));
    SQD_TRY(tl.beg.reserve((size_t)tl.nch * tl.npad * 4 + 64));
    SQD_TRY(tl.cnt.reserve((size_t)tl.nch * tl.npad * 4 + 64));
    SQD_TRY(tl.off.reserve((size_t)(tl.cap + TGRAN) * 4 + 64));
    SQD_TRY(tl.val.reserve((size_t)(tl.cap + TGRAN) * 8 + 64));
    SQD_HIP_CHECK(hipMemcpyAsync(tl.rowbase.p, tl.h_rowbase.data(), (size_t)(t.n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    SQD_HIP_CHECK(hipMemsetAsync(tl.beg.p, 0, (size_t)tl.nch * tl.npad * 4, c->stream));
    SQD_HIP_CHECK(hipMemsetAsync(tl.cnt.p, 0, (size_t)tl.nch * tl.npad * 4, c->stream));
    tb.n[sp] = t.n;
    tb.nch[sp] = tl.nch;
    tb.npad[sp] = tl.npad;
    tb.s_ptr[sp] = t.s_ptr.as<int64_t>();
    tb.d_ptr[sp] = t.d_ptr.as<int64_t>();
    tb.s_rec[sp] = t.s_rec.as<SRec>();
    tb.s_val[sp] = t.s_val.as<double>();
    tb.d_src[sp] = t.d_src.as<uint32_t>();
    tb.d_val[sp] = t.d_val.as<double>();
    tb.rowbase[sp] = tl.rowbase.as<int64_t>();
    tb.beg[sp] = tl.beg.as<uint32_t>();
    tb.cnt[sp] = tl.cnt.as<uint32_t>();
    tb.off[sp] = tl.off.as<uint32_t>();
    tb.val[sp] = tl.val.as<double>();
  }
  hipLaunchKernelGGL(k_spmm_tile_build, dim3((unsigned)((maxn + 3) / 4), 2), dim3(256), 0, c->stream, tb);
  SQD_HIP_CHECK(hipGetLastError());
  return SQD_OK;
}


  } else if (s->tiled) {
    TiledArgs tg;
    unsigned gxt = 1;
    for (int sp = 0; sp < 2; ++sp) {
      const SpmmTiles& tl = s->tiles[sp];
      tg.beg[sp] = tl.beg.as<uint32_t>();
      tg.cnt[sp] = tl.cnt.as<uint32_t>();
      tg.off[sp] = tl.off.as<uint32_t>();
      tg.val[sp] = tl.val.as<double>();
      tg.n[sp] = sp ? nb : na;
      tg.m[sp] = sp ? na : nb;
      tg.npad[sp] = tl.npad;
      tg.nch[sp] = tl.nch;
      tg.ntb[sp] = (unsigned)((tg.n[sp] + TB - 1) / TB);
      tg.npanels[sp] = (unsigned)((tg.m[sp] + TPANEL - 1) / TPANEL);
      const unsigned blocks = 8u * ((tg.npanels[sp] + 7u) / 8u) * tg.ntb[sp];
      gxt = blocks > gxt ? blocks : gxt;
    }
    tg.in[0] = d_c;
    tg.in[1] = s->ct.as<double>();
    tg.out[0] = c->gdense.as<double>();
    tg.out[1] = s->g2t.as<double>();
    tg.stop = c->sigma_stop;
    tg.vec_index = vidx;
    tg.in_stride = in_stride;
    constexpr size_t shmem = (size_t)TS * TPANEL * 8 + (size_t)(TTHREADS / 64) * TW * TSL * 12;
    static std::atomic<bool> granted[64];
    if (!granted[c->device & 63].load(std::memory_order_relaxed)) {
      SQD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spmm_tiled), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)shmem));
      granted[c->device & 63].store(true, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(k_spmm_tiled, dim3(gxt, 2), dim3(TTHREADS), shmem, c->stream, tg);

#endif
