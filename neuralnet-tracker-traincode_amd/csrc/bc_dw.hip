// Depthwise 3x3 (pad 1, stride 1|2) forward and data-gradient kernels of the bf16-COMPUTE path (bc_common.h).
//
// Same tiling as dwconv_tiled.hip - a workgroup owns (one image) x (a band of rows, full width) x (one channel block), stages the
// stencil operand ONCE in LDS already transformed, persistent over the tiles of one block, carry of the shared rows from band to band in
// the forward - with the units of the bf16 layout: a channel block is 64 channels, a pixel of it is ONE 128-byte line in global memory
// AND in LDS (the tile is kept in bf16: 8 channels per 16-byte chunk, a lane owns one chunk of a pixel), the taps are read with
// ds_read_b128, widened and accumulated in fp32 (packed two-channel fmas).  Byte for byte the access pattern of the fp32 kernels'
// 32-channel slabs, with twice the channels per byte.  C = 32 (the first block): a 32-channel block, 64-byte pixels.
// BatchNorm maps in the one-fma forms (bc_common.h); partial sums in fp32 (the stored values carry 8 bits).
#include <type_traits>

#include "bc_common.h"

// Timing-only bits of the backward kernel (experiment builds: tools/exp/build_variants.sh ... "-DTTK_BC_DBG=<bits>"; wrong results):
//   1 no tap arithmetic (G and the weight-gradient accumulators)   2 no tap LDS reads   4 no store of g_prev
//   8 no statistics                                                 16 no second-phase global loads
//   32 no second phase at all                                       64 no staging phase (loads, BatchNorm-backward map, LDS stores)
//   256 tap operands not widened (forward and backward)
//   128 no weight-gradient accumulation (72 registers less: occupancy probes with TTK_BC_BLOCK / TTK_BC_WPE)
#ifndef TTK_BC_DBG
#define TTK_BC_DBG 0
#endif
#ifndef TTK_BC_STAGGER
#define TTK_BC_STAGGER 0
#endif
// threads per workgroup / workgroups per CU of the backward kernel (experiment builds: occupancy probes; the product is 256 x 2)
#ifndef TTK_BC_BLOCK
#define TTK_BC_BLOCK 256
#endif
#ifndef TTK_BC_WPE
#define TTK_BC_WPE 2
#endif
// ring depths of the backward kernel (experiment builds may override them)
#ifndef TTK_BC_RING_STAGE
#define TTK_BC_RING_STAGE 4
#endif
#ifndef TTK_BC_RING_LEAN
#define TTK_BC_RING_LEAN 6
#endif
#ifndef TTK_BC_RING_FULL
#define TTK_BC_RING_FULL 2
#endif

namespace ttk {
namespace bc {

constexpr int kBlock = TTK_BC_BLOCK;  // (shadows ttk::kBlock inside this namespace)
constexpr int kWvs = kBlock / kWave;

constexpr int kPixBudgetFwd = 560;  // 128-byte pixels per LDS stage: 70 KB -> 2 workgroups per CU (three, with 168 registers, spilled around the load ring)
constexpr int kPixBudgetBwd = 512;  // 64 KB + 13 KB of taps / reduction scratch / constants: TWO workgroups per CU fit the 160 KB (560 did not: one per CU, one wave per SIMD); 72 weight-gradient accumulators per lane: <= 256 registers
constexpr int kWgsFwd = 2, kWgsBwd = 2;
constexpr int kCarryRegs = 5;       // 16-byte registers per thread that hand the shared rows from one band to the next
constexpr int kColTileMinW = 48, kColTile = 17;

__host__ __device__ constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }

struct Tiling {
  int SL, R, nbands, nslabs, grid, rows, NI, stage_rows, NCT, TW, carry, pix_budget;
};
static int band_rows(int Hgrid, int Wstage, int stride, bool backward, int budget) {
  const int stage_rows = budget / (Wstage + 2);
  int R;
  if (!backward) R = (stage_rows - 3) / stride + 1;
  else R = (stride == 1) ? stage_rows - 2 : 2 * (stage_rows - 2);
  if (R < 1) R = 1;
  if (R > Hgrid) R = Hgrid;
  return R;
}
static Tiling tiling(int B, int H, int W, int C, int stride, bool backward) {
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  Tiling t;
  t.SL = cbw(C);
  const int KQ = t.SL / 8;
  t.pix_budget = (backward ? kPixBudgetBwd : kPixBudgetFwd) * 64 / t.SL;
  t.NCT = 1;
  t.TW = Wo;
  t.carry = !backward && (3 - stride) * (W + 2) * KQ <= kCarryRegs * kBlock;
  if (!t.carry && stride == 1 && W >= kColTileMinW && kColTile < W) {
    t.NCT = (W + kColTile - 1) / kColTile;
    t.TW = (W + t.NCT - 1) / t.NCT;
  }
  const int Wtile = t.NCT > 1 ? t.TW : (backward ? Wo : W);
  t.R = backward ? band_rows(H, Wtile, stride, true, t.pix_budget) : band_rows(Ho, Wtile, stride, false, t.pix_budget);
  t.nbands = ((backward ? H : Ho) + t.R - 1) / t.R;
  t.nslabs = C / t.SL;
  t.stage_rows = backward ? (stride == 1 ? t.R + 2 : t.R / 2 + 2) : (t.R - 1) * stride + 3;
  t.NI = 1;
  if (t.nbands == 1) t.carry = 0;
  if (t.nbands == 1 && t.NCT == 1) {
    const int per_image = t.stage_rows * ((backward ? Wo : W) + 2);
    t.NI = t.pix_budget / per_image;
    if (t.NI > B) t.NI = B;
    if (t.NI < 1) t.NI = 1;
  }
  const int64_t tiles_per_slab = (int64_t)((B + t.NI - 1) / t.NI) * t.nbands * t.NCT;
  int64_t rows = 256 * (backward ? kWgsBwd : kWgsFwd) / t.nslabs;
  if (rows < 1) rows = 1;
  if (rows > tiles_per_slab) rows = tiles_per_slab;
  t.rows = (int)rows;
  t.grid = t.rows * t.nslabs;
  return t;
}

struct TileDiv {  // exact n / d for the tile-local indices (dwconv_tiled.hip)
  float inv;
  unsigned d;
  __device__ __forceinline__ explicit TileDiv(unsigned d_) : inv(1.0f / (float)d_), d(d_) {}
  __device__ __forceinline__ unsigned div(unsigned n) const { return (unsigned)(((float)n + 0.5f) * inv); }
};

// timing-only (TTK_BC_DBG & 256): the tap operands are NOT widened (what an fp32 LDS tile would save in the tap loops; wrong results)
__device__ __forceinline__ void unpack_tap(uint4 u, f2 (&v)[4]) {
  if (TTK_BC_DBG & 256) {
    v[0] = f2{__uint_as_float(u.x), __uint_as_float(u.y)}; v[1] = f2{__uint_as_float(u.z), __uint_as_float(u.w)};
    v[2] = f2{__uint_as_float(u.x), __uint_as_float(u.z)}; v[3] = f2{__uint_as_float(u.y), __uint_as_float(u.w)};
  } else {
    unpack_f2(u, v);
  }
}
// per-channel sums of the workgroup -> part_row[0][c], part_row[1][c] (fixed wave order); lanes KQ apart own the same chunk
template <int SL>
__device__ __forceinline__ void slab_partials(f2 (&s1)[4], f2 (&s2)[4], int q, int C, int c_slab, float* part_row, float* red /* [4][2][SL] */) {
  constexpr int KQ = SL / 8;
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int off = KQ; off < kWave; off <<= 1) {
      s1[k].x += __shfl_xor(s1[k].x, off); s1[k].y += __shfl_xor(s1[k].y, off);
      s2[k].x += __shfl_xor(s2[k].x, off); s2[k].y += __shfl_xor(s2[k].y, off);
    }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane < KQ) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      red[(wv * 2 + 0) * SL + 8 * q + 2 * k] = s1[k].x; red[(wv * 2 + 0) * SL + 8 * q + 2 * k + 1] = s1[k].y;
      red[(wv * 2 + 1) * SL + 8 * q + 2 * k] = s2[k].x; red[(wv * 2 + 1) * SL + 8 * q + 2 * k + 1] = s2[k].y;
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * SL) {
    const int which = threadIdx.x / SL, c = threadIdx.x % SL;
    float a = 0.f;
    for (int w = 0; w < kBlock / kWave; ++w) a += red[(w * 2 + which) * SL + c];
    part_row[(size_t)which * C + c_slab + c] = a;
  }
}

// filter taps of the slab into LDS: wt[tap][SL]
template <int SL>
__device__ __forceinline__ void stage_taps(float* wt, const float* __restrict__ w, int c_slab) {
  for (int i = threadIdx.x; i < 9 * SL; i += kBlock) {
    const int t = i / SL, c = i % SL;
    wt[i] = w[(size_t)(c_slab + c) * 9 + t];
  }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int S, bool SKIP, int SL, bool CARRY>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(TTK_BC_WPE, TTK_BC_WPE)))
bc_dw_fwd_k(const bf16_t* __restrict__ yprev, const float* __restrict__ bn_prev, const bf16_t* __restrict__ skip_prev, bf16_t* __restrict__ a_out,
            const float* __restrict__ w, bf16_t* __restrict__ y, float* __restrict__ part, const float* __restrict__ pivot, int B, int H, int W, int C,
            int Ho, int Wo, int R, int nbands, int nslabs, int NI_, int NCT_, int TW, int stage_pix) {
  extern __shared__ uint4 lds[];  // [stage_pix][KQ] chunks | wt[9][SL] floats | red[kWvs][2][SL] floats
  constexpr int KQ = SL / 8, kPixSlots = kBlock / KQ, kQs = ilog2(KQ), cshift = ilog2(SL);
  float* wt = reinterpret_cast<float*>(lds + (size_t)stage_pix * KQ);
  float* red = wt + 9 * SL;
  const int tid = threadIdx.x, q = tid & (KQ - 1), slot = tid >> kQs;
  const int slab = blockIdx.x % nslabs, c0 = slab * SL + 8 * q;
  stage_taps<SL>(wt, w, slab * SL);
  f2 sc[4], sh[4], pv[4];
  {
    f2 mu[4], be[4];
    ld8(bn_prev + TTK_BN_SCALE * C + c0, sc);
    ld8(bn_prev + TTK_BN_MEAN * C + c0, mu);
    ld8(bn_prev + TTK_BN_BETA * C + c0, be);
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[k] = fma2(-sc[k], mu[k], be[k]);
    if (pivot) ld8(pivot + c0, pv);
    else {
#pragma unroll
      for (int k = 0; k < 4; ++k) pv[k] = f2{0.f, 0.f};
    }
  }
  f2 s1[4], s2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) s1[k] = s2[k] = f2{0.f, 0.f};
  const int NI = CARRY ? 1 : NI_, NCT = CARRY ? 1 : NCT_;
  constexpr bool carry = CARRY;
  const unsigned tiles = (unsigned)((B + NI - 1) / NI) * nbands * NCT;
  const unsigned wgs = gridDim.x / nslabs, wg = blockIdx.x / nslabs;
  const unsigned t_begin = carry ? (unsigned)((uint64_t)tiles * wg / wgs) : wg, t_end = carry ? (unsigned)((uint64_t)tiles * (wg + 1) / wgs) : tiles;
  const unsigned t_step = carry ? 1u : wgs;
  int prev_img = -1, prev_band = -2, prev_nrows = 0;
  constexpr int OV = 3 - S;
  __syncthreads();  // taps staged
  if (TTK_BC_STAGGER && blockIdx.x >= gridDim.x / 2 && tiles >= 4u * wgs)
    for (int i = 0; i < TTK_BC_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  for (unsigned t = t_begin; t < t_end; t += t_step) {
    const unsigned tb = t / (unsigned)NCT, ti = tb / (unsigned)nbands;
    const int ct = (int)(t - tb * NCT), band = (int)(tb - ti * nbands), n0 = (int)ti * NI;
    const int nimg = min(NI, B - n0);
    const int cx0 = ct * TW, tw = NCT > 1 ? min(TW, Wo - cx0) : Wo;
    const int Wp = NCT > 1 ? tw + 2 : W + 2;
    const int o0 = band * R, o1 = min(o0 + R, Ho);
    const int i0 = o0 * S - 1;
    const int nrows = (o1 - 1 - o0) * S + 3;
    const unsigned PI = (unsigned)(nrows * Wp);
    const TileDiv dPI(PI), dWp((unsigned)Wp), dtw((unsigned)tw);
    const size_t tin = ((size_t)slab * B * H * W + (size_t)n0 * H * W) * SL, tout = ((size_t)slab * B * Ho * Wo + (size_t)n0 * Ho * Wo) * SL;
    const bf16_t* ytile = yprev + tin;
    const bf16_t* sktile = SKIP ? skip_prev + tin : nullptr;
    bf16_t* aotile = a_out ? a_out + tin : nullptr;
    bf16_t* youttile = y + tout;
    const int ov = (carry && (int)ti == prev_img && band == prev_band + 1) ? OV : 0;
    const int own_hi = (carry && t + 1 < t_end && band + 1 < nbands) ? i0 + nrows : o1;
    u32x4 cr[kCarryRegs];
    const int ncopy = ov * Wp * KQ;
    if (ov) {
      const u32x4* src = reinterpret_cast<const u32x4*>(lds + (size_t)(prev_nrows - OV) * Wp * KQ);
#pragma unroll
      for (int u = 0; u < kCarryRegs; ++u)
        if (tid + u * kBlock < ncopy) cr[u] = src[tid + u * kBlock];
    }
    prev_img = (int)ti; prev_band = band; prev_nrows = nrows;
    __syncthreads();
    if (ov) {
#pragma unroll
      for (int u = 0; u < kCarryRegs; ++u)
        if (tid + u * kBlock < ncopy) reinterpret_cast<u32x4*>(lds)[tid + u * kBlock] = cr[u];
    }
    // ---- stage a_in = relu(scale*y + shift (+ skip)) as bf16, zero outside the image
    const int nstage = nimg * ((int)PI - ov * Wp) * KQ;
    const unsigned ovpix = (unsigned)(ov * Wp);
    {
      // a ring of register slots: a slot is transformed and stored, then refilled at once, so RING elements' loads are in flight per thread all
      // the time (bytes in flight per CU are what this phase's rate follows); every load is issued unconditionally - clamped address - so
      // that hipcc's counted vmcnt waits stay exact
      constexpr int U = 1, RING = SKIP ? 4 : 8;  // eight 16-byte loads in flight per thread
      struct Stg { u32x4 y[U], sk[SKIP ? U : 1]; unsigned off[U]; int px[U]; bool in[U], own[U]; };
      auto issue = [&](int e, Stg& T) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ee = e + u * kBlock;
          const unsigned pxa = ((unsigned)ee >> kQs) + ovpix;
          const unsigned img = NI > 1 ? dPI.div(pxa) : 0u, px = NI > 1 ? pxa - __umul24(img, PI) : pxa;
          const unsigned prow = dWp.div(px);
          const int col = (int)(px - __umul24(prow, (unsigned)Wp)) - 1 + cx0, row = i0 + (int)prow;
          T.px[u] = (int)pxa;
          T.own[u] = row >= o0 && row < own_hi && col >= cx0 && col < cx0 + tw;
          T.in[u] = ee < nstage && row >= 0 && row < H && col >= 0 && col < W;
          T.off[u] = T.in[u] ? ((__umul24(__umul24(img, (unsigned)H) + (unsigned)row, (unsigned)W) + (unsigned)col) << cshift) + 8 * q : 8u * q;
          T.y[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ytile + T.off[u]));
          if constexpr (SKIP) T.sk[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sktile + T.off[u]));
        }
      };
      auto finish = [&](int e, const Stg& T) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ee = e + u * kBlock;
          if (ee >= nstage) break;
          uint4 a = make_uint4(0, 0, 0, 0);
          if (T.in[u]) {
            f2 v[4];
            unpack_f2(make_uint4(T.y[u].x, T.y[u].y, T.y[u].z, T.y[u].w), v);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fma2(sc[k], v[k], sh[k]);
            if constexpr (SKIP) {
              f2 sv[4];
              unpack_f2(make_uint4(T.sk[u].x, T.sk[u].y, T.sk[u].z, T.sk[u].w), sv);
#pragma unroll
              for (int k = 0; k < 4; ++k) v[k] += sv[k];
            }
            a = pack_f2(v);
            a.x = relu_pk(a.x); a.y = relu_pk(a.y); a.z = relu_pk(a.z); a.w = relu_pk(a.w);
            if (S == 1 && a_out && T.own[u]) st16(aotile + T.off[u], a);
          }
          lds[(size_t)T.px[u] * KQ + q] = a;
        }
      };
      Stg ring[RING];
#pragma unroll
      for (int j = 0; j < RING; ++j) issue(tid + j * kBlock, ring[j]);
      for (int e = tid; e < nstage; e += RING * kBlock) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
          finish(e + j * kBlock, ring[j]);
          issue(e + (j + RING) * kBlock, ring[j]);
        }
      }
    }
    __syncthreads();
    // ---- stencil
    const unsigned npix1 = (unsigned)((o1 - o0) * tw);
    const int npix = nimg * (int)npix1;
    const TileDiv dnp(npix1);
    for (int p = slot; p < npix; p += kPixSlots) {
      const unsigned img = NI > 1 ? dnp.div((unsigned)p) : 0u, pp = NI > 1 ? (unsigned)p - __umul24(img, npix1) : (unsigned)p;
      const unsigned prow = dtw.div(pp);
      const int ho = o0 + (int)prow, wl = (int)(pp - __umul24(prow, (unsigned)tw)), wo = cx0 + wl;
      const uint4* base = lds + ((__umul24(img, PI) + __umul24(prow * S, (unsigned)Wp) + (unsigned)(wl * S)) << kQs) + q;
      f2 acc[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = f2{0.f, 0.f};
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          f2 a[4], wv[4];
          unpack_tap(base[(kh * Wp + kw) * KQ], a);
          ld8(wt + (kh * 3 + kw) * SL + 8 * q, wv);
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[k] = fma2(a[k], wv[k], acc[k]);
        }
      const uint4 o = pack_f2(acc);
      st16(youttile + ((__umul24(__umul24(img, (unsigned)Ho) + (unsigned)ho, (unsigned)Wo) + (unsigned)wo) << cshift) + 8 * q, o);
      f2 rr[4];
      unpack_f2(o, rr);  // statistics of what is stored
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const f2 d = rr[k] - pv[k];
        s1[k] += d;
        s2[k] = fma2(d, d, s2[k]);
      }
    }
  }
  if (part) slab_partials<SL>(s1, s2, q, C, slab * SL, part + (size_t)(blockIdx.x / nslabs) * 2 * C, red);
}

// ---------------------------------------------------------------------------------------------
// data gradient (+ fused weight gradient)
// ---------------------------------------------------------------------------------------------
template <int S, int SL, bool LEAN>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(TTK_BC_WPE, TTK_BC_WPE)))
bc_dw_bwd_k(const bf16_t* __restrict__ g_dw, const bf16_t* __restrict__ y_dw, const float* __restrict__ bn_dw, const float* __restrict__ w,
            const bf16_t* __restrict__ skip_grad, const bf16_t* __restrict__ yprev, const float* __restrict__ bn_prev, const bf16_t* __restrict__ skip_prev,
            const bf16_t* __restrict__ a_in, bf16_t* __restrict__ g_prev, float* __restrict__ part, float* __restrict__ dwgrad, float* __restrict__ dw_partial,
            int B, int H, int W, int C, int Ho, int Wo, int R, int nbands, int nslabs, int stage_pix, int NI, int NCT, int TW) {
  extern __shared__ uint4 lds[];  // dy[stage_pix][KQ] chunks | wt[9][SL] | red[4][9][SL] | cst[6][SL]
  constexpr int KQ = SL / 8, kPixSlots = kBlock / KQ, kQs = ilog2(KQ), cshift = ilog2(SL);
  float* wt = reinterpret_cast<float*>(lds + (size_t)stage_pix * KQ);
  float* red = wt + 9 * SL;
  float* cst = red + kWvs * 9 * SL;
  const int tid = threadIdx.x, q = tid & (KQ - 1), slot = tid >> kQs;
  const int slab = blockIdx.x % nslabs, c0 = slab * SL + 8 * q;
  stage_taps<SL>(wt, w, slab * SL);
  // Per-channel constants in LDS, read at the start of the phase that uses them (the two phases never hold both sets in registers):
  //   [0] psc [1] psh [2] pmu  the producer's forward map a = psc*y + psh (mask / recomputed block input) and its mean (second partial sum)
  //   [3] ga  [4] gb  [5] gc   BatchNorm-backward map of this block's depthwise output: dy = ga*g + gb*y + gc
  if (tid < SL) {
    const int c = slab * SL + tid;
    const float sc_ = bn_prev[TTK_BN_SCALE * C + c], mu_ = bn_prev[TTK_BN_MEAN * C + c];
    const float ga_ = bn_dw[TTK_BN_GA * C + c], gb_ = bn_dw[TTK_BN_GB * C + c];
    cst[0 * SL + tid] = sc_;
    cst[1 * SL + tid] = fmaf(-sc_, mu_, bn_prev[TTK_BN_BETA * C + c]);
    cst[2 * SL + tid] = mu_;
    cst[3 * SL + tid] = ga_;
    cst[4 * SL + tid] = gb_;
    cst[5 * SL + tid] = -ga_ * bn_dw[TTK_BN_GMEAN * C + c] - gb_ * bn_dw[TTK_BN_MEAN * C + c];
  }
  f2 s1[4], s2[4], wacc[9][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) s1[k] = s2[k] = f2{0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) wacc[t][k] = f2{0.f, 0.f};
  const unsigned tiles = (unsigned)((B + NI - 1) / NI) * nbands * NCT;
  if (TTK_BC_STAGGER && blockIdx.x >= gridDim.x / 2 && tiles >= 4u * (gridDim.x / nslabs))
    for (int i = 0; i < TTK_BC_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
  for (unsigned t = blockIdx.x / nslabs; t < tiles; t += gridDim.x / nslabs) {
    const unsigned tb = t / (unsigned)NCT, ti = tb / (unsigned)nbands;
    const int ct = (int)(t - tb * NCT), band = (int)(tb - ti * nbands), n0 = (int)ti * NI;
    const int nimg = min(NI, B - n0);
    const int cx0 = ct * TW, tw = NCT > 1 ? min(TW, W - cx0) : W;
    const int Wp = NCT > 1 ? tw + 2 : Wo + 2;
    const int r0 = band * R, r1 = min(r0 + R, H);
    // staged dy rows ho_lo .. ho_hi: every row a tap of this band can reach, INCLUDING the rows just outside the image (staged as zeros, like
    // the two padding columns), so that the tap loop below has no bounds checks and its LDS reads can all be issued up front
    const int ho_lo = S == 1 ? r0 - 1 : r0 / 2;
    const int ho_hi = r1 / S;
    const int nrows = ho_hi - ho_lo + 1;
    const unsigned PI = (unsigned)(nrows * Wp);
    const TileDiv dPI(PI), dWp((unsigned)Wp), dtw((unsigned)tw);
    const size_t tdy = ((size_t)slab * B * Ho * Wo + (size_t)n0 * Ho * Wo) * SL, tin = ((size_t)slab * B * H * W + (size_t)n0 * H * W) * SL;
    const bf16_t* gtile = g_dw + tdy;
    const bf16_t* ydtile = y_dw + tdy;
    const bf16_t* yptile = yprev + tin;
    const bf16_t* aitile = a_in ? a_in + tin : nullptr;
    const bf16_t* sktile = skip_prev ? skip_prev + tin : nullptr;
    const bf16_t* sgtile = skip_grad ? skip_grad + tin : nullptr;
    bf16_t* gptile = g_prev + tin;
    __syncthreads();
    // ---- stage dy (bf16), zeros outside.  A ring of register slots: a slot is transformed and stored, then refilled at once (every load is
    // issued unconditionally - clamped address - so that the counted vmcnt waits stay exact).
    const int nstage = nimg * (int)PI * KQ;
    {
      constexpr int U = 1, RING = TTK_BC_RING_STAGE;  // 2 x RING 16-byte loads in flight per thread
      f2 ga[4], gb[4], gc[4];
      ld8(cst + 3 * SL + 8 * q, ga);
      ld8(cst + 4 * SL + 8 * q, gb);
      ld8(cst + 5 * SL + 8 * q, gc);
      struct Stg { u32x4 g[U], y[U]; bool in[U]; };
      auto issue = [&](int e, Stg& T) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ee = e + u * kBlock;
          const unsigned pxa = (unsigned)ee >> kQs;
          const unsigned img = NI > 1 ? dPI.div(pxa) : 0u, px = NI > 1 ? pxa - __umul24(img, PI) : pxa;
          const unsigned prow = dWp.div(px);
          const int col = (int)(px - __umul24(prow, (unsigned)Wp)) - 1 + cx0, row = ho_lo + (int)prow;
          T.in[u] = ee < nstage && col >= 0 && col < Wo && row >= 0 && row < Ho;
          const unsigned off = T.in[u] ? ((__umul24(__umul24(img, (unsigned)Ho) + (unsigned)row, (unsigned)Wo) + (unsigned)col) << cshift) + 8 * q : 8u * q;
          T.g[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(gtile + off));
          T.y[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ydtile + off));
        }
      };
      auto finish = [&](int e, const Stg& T) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int ee = e + u * kBlock;
          if (ee >= nstage) break;
          uint4 d = make_uint4(0, 0, 0, 0);
          if (T.in[u]) {
            f2 gg[4], yy[4];
            unpack_f2(make_uint4(T.g[u].x, T.g[u].y, T.g[u].z, T.g[u].w), gg);
            unpack_f2(make_uint4(T.y[u].x, T.y[u].y, T.y[u].z, T.y[u].w), yy);
#pragma unroll
            for (int k = 0; k < 4; ++k) gg[k] = fma2(ga[k], gg[k], fma2(gb[k], yy[k], gc[k]));
            d = pack_f2(gg);
          }
          lds[(size_t)(ee >> kQs) * KQ + q] = d;
        }
      };
      Stg ring[RING];
      if (!(TTK_BC_DBG & 64)) {
#pragma unroll
        for (int j = 0; j < RING; ++j) issue(tid + j * kBlock, ring[j]);
        for (int e = tid; e < nstage; e += RING * kBlock) {
#pragma unroll
          for (int j = 0; j < RING; ++j) {
            finish(e + j * kBlock, ring[j]);
            issue(e + (j + RING) * kBlock, ring[j]);
          }
        }
      }
    }
    __syncthreads();
    // ---- the pixels of the band.  Stride 2: class by class of (row parity, column parity) - a pixel of a class is reached by a fixed set of
    // 1, 2, 2 or 4 taps, so the lanes of a wave do the same work (enumerated in one loop over all pixels every wave ran all nine tap
    // bodies under masks: 266 vector instructions per pixel where ~2.25 taps do work).
    struct Item { u32x4 yp, raw, sg; int pj; unsigned off; int lb; };  // off: element offset of the pixel; lb: LDS chunk index of its first tap (decoded once, at issue)
    f2 psc[4], psh[4], pmu[4];
    ld8(cst + 0 * SL + 8 * q, psc);
    ld8(cst + 1 * SL + 8 * q, psh);
    ld8(cst + 2 * SL + 8 * q, pmu);
    auto run = [&](auto phc, auto pwc) {
      constexpr int PH = decltype(phc)::value, PW = decltype(pwc)::value;  // -1: every pixel (stride 1)
      const int fr = S == 1 ? r0 : r0 + ((PH - r0) & 1), fc = S == 1 ? cx0 : cx0 + ((PW - cx0) & 1);
      const int nr = S == 1 ? r1 - r0 : (r1 - fr + 1) / 2, nc = S == 1 ? tw : (cx0 + tw - fc + 1) / 2;
      if (nr <= 0 || nc <= 0) return;
      const unsigned npc = (unsigned)(nr * nc);
      const int npix = nimg * (int)npc;
      const TileDiv dnp(npc), dnc((unsigned)nc);
      // item pj of the class -> image of the tile, pixel, element offset (absent items: the class's first pixel, a valid address)
      auto decode = [&](int pj, unsigned& img, int& hi, int& wi, unsigned& off) {
        const unsigned pq = pj < npix ? (unsigned)pj : 0u;
        img = NI > 1 ? dnp.div(pq) : 0u;
        const unsigned pp = NI > 1 ? pq - __umul24(img, npc) : pq;
        const unsigned pr = dnc.div(pp);
        hi = fr + (int)pr * (S == 1 ? 1 : 2);
        wi = fc + (int)(pp - __umul24(pr, (unsigned)nc)) * (S == 1 ? 1 : 2);
        off = ((__umul24(__umul24(img, (unsigned)H) + (unsigned)hi, (unsigned)W) + (unsigned)wi) << cshift) + 8 * q;
      };
      auto issue = [&](int it, Item& I) {
        I.pj = slot + it * kPixSlots;
        unsigned img, off;
        int hi, wi;
        decode(I.pj, img, hi, wi, off);
        I.off = off;
        {  // tap (kh, kw) of this pixel reads chunk lb - ((kh - kh0) / S) * Wp - (kw - kw0) / S  (kh0, kw0: the class's first tap; stride 1: 0)
          constexpr int kh0 = S == 1 ? 0 : (PH + 1) & 1, kw0 = S == 1 ? 0 : (PW + 1) & 1;
          I.lb = (int)__umul24(img, PI) + __mul24((hi + 1 - kh0) / S - ho_lo, Wp) + (wi + 1 - kw0) / S - cx0 + 1;
        }
        if (TTK_BC_DBG & 16) { I.yp = u32x4{(unsigned)off, 1u, 2u, 3u}; I.raw = I.yp; I.sg = I.yp; return; }
        I.yp = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(yptile + off));
        if (!LEAN) {
          if (a_in) I.raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(aitile + off));
          else if (skip_prev) I.raw = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sktile + off));
          if (skip_grad) I.sg = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(sgtile + off));
        }
      };
      auto process = [&](const Item& I) {
        if (I.pj >= npix) return;
        // kCache: the pixel's offsets come from the item (decoded at issue); the stride-1 kernel with residual operands is out of registers
        // and decodes again (measured: caching there costs 2 us on the 9 x 9 layers, elsewhere it saves 2 - 8 %)
        constexpr bool kCache = LEAN || S == 2;
        unsigned off = I.off;
        int lb = I.lb;
        if constexpr (!kCache) {
          unsigned img;
          int hi, wi;
          decode(I.pj, img, hi, wi, off);
          lb = (int)__umul24(img, PI) + __mul24(hi + 1 - ho_lo, Wp) + wi + 1 - cx0 + 1;
        }
        const uint4* dyimg = lds + q;
        f2 yp[4], a[4];
        unpack_f2(make_uint4(I.yp.x, I.yp.y, I.yp.z, I.yp.w), yp);
        if (!LEAN && a_in) {
          unpack_f2(make_uint4(I.raw.x, I.raw.y, I.raw.z, I.raw.w), a);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) a[k] = fma2(psc[k], yp[k], psh[k]);
          if (!LEAN && skip_prev) {
            f2 sk[4];
            unpack_f2(make_uint4(I.raw.x, I.raw.y, I.raw.z, I.raw.w), sk);
#pragma unroll
            for (int k = 0; k < 4; ++k) a[k] += sk[k];
          }
          // the block input as the forward kernel formed it: rounded to bf16, then relu
          uint4 ar = pack_f2(a);
          ar.x = relu_pk(ar.x); ar.y = relu_pk(ar.y); ar.z = relu_pk(ar.z); ar.w = relu_pk(ar.w);
          unpack_f2(ar, a);
        }
        // the taps of this pixel (stride 2: those of its parity class - compile time): all their LDS reads first, then the arithmetic
        uint4 dv[9];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            if (S == 2 && (((PH + 1 - kh) & 1) || ((PW + 1 - kw) & 1))) continue;
            // (stride 2: the taps of the class; stride 1: rows -1 .. H, columns -1 .. W - the zero border of the stage)
            constexpr int kh0 = S == 1 ? 0 : (PH + 1) & 1, kw0 = S == 1 ? 0 : (PW + 1) & 1;
            const int idx = lb - __mul24((kh - kh0) / S, Wp) - (kw - kw0) / S;
            dv[kh * 3 + kw] = (TTK_BC_DBG & 2) ? make_uint4(idx, off, kh, kw) : dyimg[idx << kQs];
          }
        f2 G[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) G[k] = f2{0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if (S == 2 && (((PH + 1 - t / 3) & 1) || ((PW + 1 - t % 3) & 1))) continue;
          if (TTK_BC_DBG & 1) { G[0].x += __uint_as_float(dv[t].x ^ dv[t].y ^ dv[t].z ^ dv[t].w); continue; }
          f2 dy[4], wv[4];
          unpack_tap(dv[t], dy);
          ld8(wt + t * SL + 8 * q, wv);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            G[k] = fma2(dy[k], wv[k], G[k]);
            if (!(TTK_BC_DBG & 128)) wacc[t][k] = fma2(dy[k], a[k], wacc[t][k]);
          }
        }
        if (!LEAN && skip_grad) {
          f2 sg[4];
          unpack_f2(make_uint4(I.sg.x, I.sg.y, I.sg.z, I.sg.w), sg);
#pragma unroll
          for (int k = 0; k < 4; ++k) G[k] += sg[k];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          G[k].x = a[k].x > 0.f ? G[k].x : 0.f;
          G[k].y = a[k].y > 0.f ? G[k].y : 0.f;
        }
        const uint4 o = pack_f2(G);
        if (!(TTK_BC_DBG & 4)) st16(gptile + off, o);
        else if (o.x == 0x12345u) st16(gptile + off, o);
        f2 gp[4];
        unpack_f2(o, gp);  // sums of what is stored
        if (TTK_BC_DBG & 8) { s1[0] += gp[0] + gp[1] + gp[2] + gp[3]; return; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          s1[k] += gp[k];
          s2[k] = fma2(gp[k], yp[k] - pmu[k], s2[k]);
        }
      };
      // a ring of items: one is processed, its slot refilled at once - RING pixels' loads (1 or 3 tensors each) in flight per thread
      constexpr int RING = LEAN ? TTK_BC_RING_LEAN : TTK_BC_RING_FULL;
      const int nit = (npix + kPixSlots - 1) / kPixSlots;
      Item ring[RING];
#pragma unroll
      for (int j = 0; j < RING; ++j) issue(j, ring[j]);
      for (int it = 0; it < nit; it += RING) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
          process(ring[j]);
          issue(it + j + RING, ring[j]);
        }
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using IA = std::integral_constant<int, -1>;
    if (TTK_BC_DBG & 32) continue;
    if constexpr (S == 1) {
      run(IA{}, IA{});
    } else {
      run(I1{}, I1{});
      run(I1{}, I0{});
      run(I0{}, I1{});
      run(I0{}, I0{});
    }
  }
  if (part) slab_partials<SL>(s1, s2, q, C, slab * SL, part + (size_t)(blockIdx.x / nslabs) * 2 * C, red);
  if (dwgrad) {
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int off = KQ; off < kWave; off <<= 1) {
          wacc[t][k].x += __shfl_xor(wacc[t][k].x, off);
          wacc[t][k].y += __shfl_xor(wacc[t][k].y, off);
        }
    __syncthreads();
    if (lane < KQ) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          red[((size_t)wv * 9 + t) * SL + 8 * q + 2 * k] = wacc[t][k].x;
          red[((size_t)wv * 9 + t) * SL + 8 * q + 2 * k + 1] = wacc[t][k].y;
        }
    }
    __syncthreads();
    for (int i = tid; i < 9 * SL; i += kBlock) {
      const int t = i / SL, c = i % SL;
      float a = 0.f;
      for (int wq = 0; wq < kBlock / kWave; ++wq) a += red[((size_t)wq * 9 + t) * SL + c];
      if (dw_partial) dw_partial[((size_t)(blockIdx.x / nslabs) * C + slab * SL + c) * 9 + t] = a;
      else atomicAdd(dwgrad + (size_t)(slab * SL + c) * 9 + t, a);
    }
  }
}

static bool shape_ok(int B, int H, int W, int C, int stride) {
  return B > 0 && H > 0 && W > 0 && W <= 256 && C >= 32 && C <= 1024 && (C & (C - 1)) == 0 && (stride == 1 || stride == 2);
}

}  // namespace bc
}  // namespace ttk

using namespace ttk;
using namespace ttk::bc;

extern "C" {

int ttk_bc_partial_rows_dw(int B, int H, int W, int C, int stride, int backward) {
  if (!shape_ok(B, H, W, C, stride)) return -1;
  return tiling(B, H, W, C, stride, backward != 0).rows;
}

int ttk_bc_dw_fwd(const void* yprev, const float* bn_prev, const void* skip_prev, void* a_out, const float* w, void* y, float* part, const float* pivot,
                  int B, int H, int W, int C, int stride, ttk_stream_t stream) {
  TTK_REQUIRE(yprev && bn_prev && w && y, "bc_dw_fwd: null pointer");
  TTK_REQUIRE(shape_ok(B, H, W, C, stride), "bc_dw_fwd: unsupported shape B=%d H=%d W=%d C=%d stride=%d", B, H, W, C, stride);
  TTK_REQUIRE(!(a_out && stride != 1), "bc_dw_fwd: a_out requires stride 1");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const Tiling t = tiling(B, H, W, C, stride, false);
  TTK_REQUIRE((int64_t)((B + t.NI - 1) / t.NI) * t.nbands * t.NCT < ((int64_t)1 << 31) && (int64_t)B * H * W * t.SL < ((int64_t)1 << 32), "bc_dw_fwd: too large for 32-bit indexing");
  const int stage_pix = t.NI * t.stage_rows * ((t.NCT > 1 ? t.TW : W) + 2);
  const size_t sm = (size_t)stage_pix * t.SL * 2 + (size_t)(9 + 2 * kWvs) * t.SL * sizeof(float);
#define TTK_BC_FWD3(S_, SK_, SL_, CY_)                                                                                                            \
  (void)allow_big_lds<bc_dw_fwd_k<S_, SK_, SL_, CY_>>();                                                                                                \
  hipLaunchKernelGGL((bc_dw_fwd_k<S_, SK_, SL_, CY_>), dim3(t.grid), dim3(bc::kBlock), sm, (hipStream_t)stream, (const bf16_t*)yprev, bn_prev,          \
                     (const bf16_t*)skip_prev, (bf16_t*)a_out, w, (bf16_t*)y, part, pivot, B, H, W, C, Ho, Wo, t.R, t.nbands, t.nslabs, t.NI, t.NCT, t.TW, stage_pix)
#define TTK_BC_FWD2(S_, SK_, SL_) do { if (t.carry) { TTK_BC_FWD3(S_, SK_, SL_, true); } else { TTK_BC_FWD3(S_, SK_, SL_, false); } } while (0)
#define TTK_BC_FWD1(S_, SK_) do { if (t.SL == 64) TTK_BC_FWD2(S_, SK_, 64); else TTK_BC_FWD2(S_, SK_, 32); } while (0)
  if (stride == 1) { if (skip_prev) TTK_BC_FWD1(1, true); else TTK_BC_FWD1(1, false); }
  else { if (skip_prev) TTK_BC_FWD1(2, true); else TTK_BC_FWD1(2, false); }
#undef TTK_BC_FWD1
#undef TTK_BC_FWD2
#undef TTK_BC_FWD3
  TTK_LAUNCH_CHECK("bc_dw_fwd");
}

int ttk_bc_dw_bwd_data(const void* g_dw, const void* y_dw, const float* bn_dw, const float* w, const void* skip_grad, const void* yprev,
                       const float* bn_prev, const void* skip_prev, const void* a_in, void* g_prev, float* part, float* dw, int dw_accumulate,
                       float* dw_partial, int B, int H, int W, int C, int stride, ttk_stream_t stream) {
  TTK_REQUIRE(g_dw && y_dw && bn_dw && w && yprev && bn_prev && g_prev, "bc_dw_bwd_data: null pointer");
  TTK_REQUIRE(shape_ok(B, H, W, C, stride), "bc_dw_bwd_data: unsupported shape");
  TTK_REQUIRE(!(skip_grad && stride != 1), "bc_dw_bwd_data: residual gradient requires stride 1");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const Tiling t = tiling(B, H, W, C, stride, true);
  TTK_REQUIRE((int64_t)((B + t.NI - 1) / t.NI) * t.nbands * t.NCT < ((int64_t)1 << 31) && (int64_t)B * H * W * t.SL < ((int64_t)1 << 32), "bc_dw_bwd_data: too large for 32-bit indexing");
  const int stage_pix = t.NI * t.stage_rows * ((t.NCT > 1 ? t.TW : Wo) + 2);
  const size_t sm = (size_t)stage_pix * t.SL * 2 + (size_t)(9 + kWvs * 9 + 6) * t.SL * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (!dw) dw_partial = nullptr;
  TTK_REQUIRE(dw_accumulate != 2 || dw_partial, "bc_dw_bwd_data: dw_accumulate = 2 needs dw_partial");
  if (dw && !dw_accumulate && !dw_partial) (void)hipMemsetAsync(dw, 0, (size_t)9 * C * sizeof(float), st);
#define TTK_BC_BWD3(S_, SL_, LEAN_)                                                                                                               \
  do {                                                                                                                                            \
    (void)allow_big_lds<bc_dw_bwd_k<S_, SL_, LEAN_>>();                                                                                                \
    hipLaunchKernelGGL((bc_dw_bwd_k<S_, SL_, LEAN_>), dim3(t.grid), dim3(bc::kBlock), sm, st, (const bf16_t*)g_dw, (const bf16_t*)y_dw, bn_dw, w,      \
                       (const bf16_t*)skip_grad, (const bf16_t*)yprev, bn_prev, (const bf16_t*)skip_prev, (const bf16_t*)a_in, (bf16_t*)g_prev, part, dw, \
                       dw_partial, B, H, W, C, Ho, Wo, t.R, t.nbands, t.nslabs, stage_pix, t.NI, t.NCT, t.TW);                                     \
  } while (0)
#define TTK_BC_BWD2(S_, SL_) do { if (lean) TTK_BC_BWD3(S_, SL_, true); else TTK_BC_BWD3(S_, SL_, false); } while (0)
#define TTK_BC_BWD1(S_) do { if (t.SL == 64) TTK_BC_BWD2(S_, 64); else TTK_BC_BWD2(S_, 32); } while (0)
  const bool lean = !a_in && !skip_prev && !skip_grad;
  if (stride == 1) TTK_BC_BWD1(1); else TTK_BC_BWD1(2);
#undef TTK_BC_BWD1
#undef TTK_BC_BWD2
#undef TTK_BC_BWD3
  // dw_accumulate == 2: the rows stay unfolded - the caller folds them beside the BatchNorm-backward finalisation (ttk_bc_bn_bwd_finalize_fold)
  if (dw_partial && dw_accumulate != 2 && !launch_fold_rows_fast(dw_partial, t.rows, (int64_t)9 * C, dw, dw_accumulate, st))
    launch_fold_partials(dw_partial, t.rows, (int64_t)9 * C, dw, dw_accumulate, st);
  TTK_LAUNCH_CHECK("bc_dw_bwd_data");
}

}  // extern "C"
