// LDS-tiled depthwise 3x3 (pad 1, stride 1|2) forward and data-gradient kernels.
//
// v1 of these kernels read every tap straight from global memory: 9 (forward) / 18 (backward)
// 16-byte loads per output float4 through L1/TA plus a 9x repeated BatchNorm transform - they ran at
// 1.5-2 TB/s, texture-addresser bound.  Here a workgroup owns a tile
//     (one image) x (a band of R rows, full width) x (a slab of 32 channels)
// stages the operand of the stencil ONCE in LDS - already transformed (forward: a_in = relu(bn(y_prev)
// (+skip)); backward: dy = ga*(g-gmean)+gb*(y-mean)) - with one zero column of padding left and right,
// and reads the 9 taps with ds_read_b128 (8 lanes x 16 B = the 128-byte channel slab of one pixel;
// 8 consecutive pixels of a row = 1 KiB contiguous: conflict-free).  In global memory the tensors are channel blocks
// [C/32][pixels][32] (ttk_common.h): the slab of a tile row is contiguous there too.
//
// Workgroups are persistent over tiles of ONE channel slab, so per-channel BatchNorm partial sums and
// the fused depthwise weight gradient accumulate in registers across tiles and leave the block once
// (one partial row + 288 float atomics per workgroup).
#include "ttk_common.h"
#ifndef TTK_DW_FWD_U
#define TTK_DW_FWD_U 6
#endif
#ifndef TTK_DW_BWD_U
#define TTK_DW_BWD_U 4
#endif
#ifndef TTK_DW_FWD_U2
#define TTK_DW_FWD_U2 6
#endif

namespace ttk {

// Channels per tile ("slab") = one channel block of the activation layout (ttk_common.h act_off): the pixels of a slab are 128 bytes
// apart, a tile row is one contiguous run.  (Over channels-last rows a workgroup touched one 128-byte piece per pixel, 4 C bytes
// apart: tools/stream_sweep.py, profiles/r03_stream_sweep.txt - 5:1 read:write mix at C = 512: 128-byte pieces 4.9 TB/s, 256-byte
// pieces 5.5, linear 6.2.  What 64-channel slabs measured there is at dw_tiling().)
#ifndef TTK_DW_LDS_PIX
#define TTK_DW_LDS_PIX 368
#endif
constexpr int kLdsPixBudget32 = TTK_DW_LDS_PIX;   // (rows) x (W+2) pixels of 128 B each: <= 47 KB -> 3 workgroups per CU (560 -> 2 per CU measured slower)
__host__ __device__ constexpr int lds_pix_budget(int SL) { return kLdsPixBudget32 * 32 / SL; }
__host__ __device__ constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
#ifndef TTK_DW_WGS_PER_CU
#define TTK_DW_WGS_PER_CU 3
#endif
constexpr int kMaxDwBlocks = 256 * TTK_DW_WGS_PER_CU;   // 3 workgroups per CU x 256 CUs: one resident wave of persistent workgroups
// staging elements per thread and iteration (forward): their loads are in flight together, and the bytes in flight per CU are
// what these kernels' throughput follows.  Six fit the 168-register budget of three workgroups per CU when the layer has no
// residual input to load beside them (the largest layers), four otherwise.
constexpr int kFwdUSkip = 4, kFwdUPlain = TTK_DW_FWD_U, kFwdUPlain2 = TTK_DW_FWD_U2;

struct DwTiling {
  int SL;                             // channels per slab: 32 | 64
  int R, nbands, nslabs, grid, rows;  // rows = partial rows = grid / nslabs
  int NI;                             // images per tile (> 1 only when one band covers the image: the 9x9 and 5x5 layers)
  int stage_rows;                     // LDS rows of one image's stage
  int NCT, TW;                        // column tiles per band and their width (stride 1, wide images); else 1, full width
  int carry;                          // a workgroup walks CONSECUTIVE bands of an image and hands their shared staged rows from band to band
};
#ifndef TTK_DW_CARRY
#define TTK_DW_CARRY 1
#endif
#ifndef TTK_DW_CARRY_BWD
#define TTK_DW_CARRY_BWD 1
#endif
constexpr int kCarryRegs = 5;  // float4 registers per thread that hand the shared rows from one band to the next

constexpr int kColTileMinW = 48;  // images at least this wide (the 65x65 layer) are tiled in columns too
constexpr int kColTile = 17;      // 19 x 19 staged pixels for 17 x 17 results: halo 1.25x instead of 1.7x for 3-row bands

// band height on the grid the kernel iterates (forward: output rows; backward: input rows)
__host__ __device__ inline int dw_band_rows(int Hgrid, int Wstage, int stride, bool backward, int SL = 32) {
  const int stage_rows = lds_pix_budget(SL) / (Wstage + 2);  // LDS rows we can afford
  int R;
  if (!backward) R = (stage_rows - 3) / stride + 1;      // needs (R-1)*S+3 input rows
  else R = (stride == 1) ? stage_rows - 2 : 2 * (stage_rows - 2);  // needs <= R/S+2 output rows
  if (R < 1) R = 1;
  if (R > Hgrid) R = Hgrid;
  return R;
}

inline DwTiling dw_tiling_sl(int B, int H, int W, int C, int stride, bool backward, int SL) {
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  DwTiling t;
  t.SL = SL;
  t.NCT = 1;
  t.TW = Wo;
  static const int col_tile = [] { const char* e = exp_env("TTK_DW_COLTILE"); return e ? atoi(e) : kColTile; }();  // (experiments)
  // Forward, images of several full-width bands (65 x 65, 33 x 33): the 3 - stride input rows two neighbouring bands share are staged
  // ONCE - a workgroup takes a contiguous run of bands and carries those rows over in LDS - so the tensor is read exactly once
  // (column tiles of 17 x 17 results staged 19 x 19 = 1.25 x; full-width bands without the carry 5 rows for 3 = 1.67 x; PMC of
  // round 3: 480 MB for 404 on the stride-1 kernel, 833 for 696 on the stride-2 65 x 65 x 64 layer).
  // Backward (round 6): the same for the staged dy rows - neighbouring bands of input rows share 2 (stride 1) or 1 (stride 2) rows of dy - with
  // full-width bands instead of the 19 x 19 column tiles of the 65-pixel layers (halo 1.25 x on g and y) and the 10-for-8-row bands of the
  // 33-pixel ones.
  static const int carry_bwd = [] { const char* e = exp_env("TTK_DW_CARRY_BWD"); return e ? atoi(e) : TTK_DW_CARRY_BWD; }();  // (experiments)
  // (stride 2, where the bands share ONE dy row of a tensor a quarter of the input's size, measured slower with the ring: 65 x 65 x 64 259 -> 268 us;
  // stride 1: 33 x 33 x 128 319 -> 284 us, 65 x 65 x 32 217 -> 190 us, profiles/r06_depthwise_backward_carry.txt)
  t.carry = backward ? (carry_bwd && stride == 1 && (Wo + 2) <= 80)
                     : (TTK_DW_CARRY && (3 - stride) * (W + 2) * (SL / 4) <= kCarryRegs * kBlock);
  if (!t.carry && stride == 1 && W >= kColTileMinW && col_tile < W) {
    t.NCT = (W + col_tile - 1) / col_tile;
    t.TW = (W + t.NCT - 1) / t.NCT;
  }
  const int Wtile = t.NCT > 1 ? t.TW : (backward ? Wo : W);
  t.R = backward ? dw_band_rows(H, Wtile, stride, true, SL) : dw_band_rows(Ho, Wtile, stride, false, SL);
  t.nbands = ((backward ? H : Ho) + t.R - 1) / t.R;
  t.nslabs = C / SL;
  t.stage_rows = backward ? (stride == 1 ? t.R + 2 : t.R / 2 + 2) : (t.R - 1) * stride + 3;
  // Small images: one tile = several whole images side by side in LDS (each with its own zero border).  A 5x5 image
  // is 49 staged pixels - a fraction of one pass of the 256 threads between two barriers; seven of them fill the
  // stage buffer, the lanes and the memory pipeline.
  t.NI = 1;
  if (t.nbands == 1) t.carry = 0;
  if (t.nbands == 1 && t.NCT == 1) {
    const int per_image = t.stage_rows * ((backward ? Wo : W) + 2);
    t.NI = lds_pix_budget(SL) / per_image;
    if (t.NI > B) t.NI = B;
    if (t.NI < 1) t.NI = 1;
  }
  int64_t tiles_per_slab = (int64_t)((B + t.NI - 1) / t.NI) * t.nbands * t.NCT;
  int64_t rows = kMaxDwBlocks / t.nslabs;
  if (rows < 1) rows = 1;
  if (rows > tiles_per_slab) rows = tiles_per_slab;
  t.rows = (int)rows;
  t.grid = t.rows * t.nslabs;
  return t;
}

inline DwTiling dw_tiling(int B, int H, int W, int C, int stride, bool backward) {
  // A slab is one 32-channel block of the activation layout.  (Round 3 also measured 64-channel slabs over the channels-last layout
  // of that time: forward unchanged, backward 5-10 % slower - half the pixels per LDS stage; the kernels keep SL as a template
  // parameter for that history, only SL = kCB is instantiated.)
  return dw_tiling_sl(B, H, W, C, stride, backward, kCB);
}

// n / d for the tile-local pixel indices (0 <= n < 2^20, 1 <= d < 2^12) in four VALU operations: (n + 0.5) / d is at least 0.5 / d
// away from every integer and the float product is off by less than 2^-22 of its value, so the truncation is exact.  The
// integer divisions these replace (~30 operations each, two to six per staged or produced pixel) made the kernels of the large
// layers VALU-bound: timing-only builds without loads, without stores or with one tap all ran within 7 % of the full kernel.
struct TileDiv {
  float inv;
  unsigned d;
  __device__ __forceinline__ explicit TileDiv(unsigned d_) : inv(1.0f / (float)d_), d(d_) {}
  __device__ __forceinline__ unsigned div(unsigned n) const { return (unsigned)(((float)n + 0.5f) * inv); }
};

struct SlabWeights {  // w[c][tap] of 4 consecutive channels
  float v[36];
  __device__ __forceinline__ void load(const float* w, int c) {
    const float4* p = reinterpret_cast<const float4*>(w + 9 * c);
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float4 q = p[i];
      v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
    }
  }
  __device__ __forceinline__ float4 tap(int t) const { return make_float4(v[t], v[9 + t], v[18 + t], v[27 + t]); }
};

// fold over the pixel slots of the block that sit in one wave: lanes SL / 4 apart own the same quad
template <int SL>
__device__ __forceinline__ float4 slab_wave_fold(float4 v) {
#pragma unroll
  for (int off = SL / 4; off < kWave; off <<= 1) {
    v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off);
    v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
  }
  return v;
}

// Per-channel sums carried in fp64: a persistent workgroup adds thousands of terms per thread and
// sum(g) cancels heavily (BatchNorm-backward makes the upstream gradient zero-mean), so fp32 running sums
// cost 1-2 digits of d(beta).  fp64 adds are free next to the memory time of these kernels.
struct D4 {
  double x, y, z, w;
  __device__ __forceinline__ void add(float4 v) { x += v.x; y += v.y; z += v.z; w += v.w; }
  __device__ __forceinline__ void addmul(float4 a, float4 b) {
    x += (double)a.x * b.x; y += (double)a.y * b.y; z += (double)a.z * b.z; w += (double)a.w * b.w;
  }
};
__device__ __forceinline__ double shfl_xor_d(double v, int off) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, off);
  hi = __shfl_xor(hi, off);
  return __hiloint2double(hi, lo);
}
template <int SL>
__device__ __forceinline__ D4 slab_wave_fold_d(D4 v) {
#pragma unroll
  for (int off = SL / 4; off < kWave; off <<= 1) {
    v.x += shfl_xor_d(v.x, off); v.y += shfl_xor_d(v.y, off);
    v.z += shfl_xor_d(v.z, off); v.w += shfl_xor_d(v.w, off);
  }
  return v;
}

// writes part_row[0][slab columns] = sum s1, part_row[1][slab columns] = sum s2 (fixed wave order)
template <int SL>
__device__ __forceinline__ void slab_partials(D4 s1, D4 s2, int q, int C, int c_slab, float* part_row, float* red_f) {
  double* red = reinterpret_cast<double*>(red_f);  // [4 waves][2][SL] doubles
  s1 = slab_wave_fold_d<SL>(s1);
  s2 = slab_wave_fold_d<SL>(s2);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane < SL / 4) {
    double* d = red + (wv * 2 + 0) * SL + 4 * q;
    d[0] = s1.x; d[1] = s1.y; d[2] = s1.z; d[3] = s1.w;
    d = red + (wv * 2 + 1) * SL + 4 * q;
    d[0] = s2.x; d[1] = s2.y; d[2] = s2.z; d[3] = s2.w;
  }
  __syncthreads();
  if (threadIdx.x < 2 * SL) {
    const int which = threadIdx.x / SL, c = threadIdx.x % SL;
    double a = 0.0;
    for (int w = 0; w < kBlock / kWave; ++w) a += red[(w * 2 + which) * SL + c];
    part_row[(size_t)which * C + c_slab + c] = (float)a;
  }
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int S, typename T, bool SKIP, int SL, bool CARRY>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(TTK_DW_WGS_PER_CU, TTK_DW_WGS_PER_CU)))
dw_fwd_tiled_k(const T* __restrict__ yprev, const float* __restrict__ bn_prev,
                                                          const T* __restrict__ skip_prev, T* __restrict__ a_out,
                                                          const float* __restrict__ w, T* __restrict__ y,
                                                          float* __restrict__ part, const float* __restrict__ pivot, int B, int H, int W, int C,
                                                          int Ho, int Wo, int R, int nbands, int nslabs, int NI_, int NCT_, int TW) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // [NI][stage_rows][tile width + 2][SL] + reduction scratch
  constexpr int kSlab = SL, kSlabQuads = SL / 4, kPixSlots = kBlock / kSlabQuads, kQs = ilog2(kSlabQuads), kPs = ilog2(SL);
  const int tid = threadIdx.x, q = tid & (kSlabQuads - 1), slot = tid >> kQs;
  const int slab = blockIdx.x % nslabs, c0 = slab * kSlab + 4 * q;
  static_assert(SL == kCB, "a slab is one channel block of the activation layout");
  constexpr int cshift = 5;  // pixels of a channel block are 32 elements apart (ttk_common.h act_off)
  SlabWeights wr;
  wr.load(w, c0);
  const BnApply4 bn = BnApply4::load(bn_prev, C, c0);
  const float4 pv = pivot ? ld4(pivot + c0) : f4(0.f);  // the partial sums are those of y - pivot (ttk.h)
  D4 s1{0.0, 0.0, 0.0, 0.0}, s2{0.0, 0.0, 0.0, 0.0};
  const int NI = CARRY ? 1 : NI_, NCT = CARRY ? 1 : NCT_;  // (carry mode: one image per tile, full-width bands - constants for the compiler)
  constexpr bool carry = CARRY;
  const unsigned tiles = (unsigned)((B + NI - 1) / NI) * nbands * NCT;  // (< 2^31: checked by the host)
  // Tile order.  Default: workgroup r of a slab takes tiles r, r + rows, ...  Carry mode (full-width bands of a several-band image, NI =
  // NCT = 1): a CONTIGUOUS run of tiles, i.e. consecutive bands of an image, whose 3 - S shared input rows go from one band's stage to
  // the next through registers instead of being read again.
  const unsigned wgs = gridDim.x / nslabs, wg = blockIdx.x / nslabs;
  const unsigned t_begin = carry ? (unsigned)((uint64_t)tiles * wg / wgs) : wg, t_end = carry ? (unsigned)((uint64_t)tiles * (wg + 1) / wgs) : tiles;
  const unsigned t_step = carry ? 1u : wgs;
  int prev_img = -1, prev_band = -2, prev_nrows = 0;
  constexpr int OV = 3 - S;  // input rows two neighbouring bands share
  for (unsigned t = t_begin; t < t_end; t += t_step) {
    const unsigned tb = t / (unsigned)NCT, ti = tb / (unsigned)nbands;
    const int ct = (int)(t - tb * NCT), band = (int)(tb - ti * nbands), n0 = (int)ti * NI;  // NI > 1 implies one tile per image
    const int nimg = min(NI, B - n0);
    const int cx0 = ct * TW, tw = NCT > 1 ? min(TW, Wo - cx0) : Wo;  // output columns [cx0, cx0 + tw) (column tiles: stride 1)
    const int Wp = NCT > 1 ? tw + 2 : W + 2;                          // staged input columns cx0-1 .. cx0+tw (or -1 .. W)
    const int o0 = band * R, o1 = min(o0 + R, Ho);
    const int i0 = o0 * S - 1;                     // first staged input row (may be -1)
    const int nrows = (o1 - 1 - o0) * S + 3;
    const unsigned PI = (unsigned)(nrows * Wp);    // staged pixels per image
    const TileDiv dPI(PI), dWp((unsigned)Wp), dtw((unsigned)tw);
    // Addresses: one 64-bit base per tile and tensor, 32-bit element offsets formed with 24-bit multiplies (v_mul_u32_u24 is a
    // full-rate instruction; the 32-bit / 64-bit integer multiplies of the size_t form run at a quarter of that rate, and at
    // ten per staged element they were a third of these VALU-bound kernels' issue slots)
    // (the bases are uniform over the workgroup - scalar registers; the lane's channel quad rides in the 32-bit offset)
    // channel block `slab` of the input / output tensor, first pixel of image n0: everything this tile touches is one contiguous run
    const size_t tin = ((size_t)slab * B * H * W + (size_t)n0 * H * W) * kCB, tout = ((size_t)slab * B * Ho * Wo + (size_t)n0 * Ho * Wo) * kCB;
    const T* ytile = yprev + tin;
    const T* sktile = SKIP ? skip_prev + tin : nullptr;
    T* aotile = a_out ? a_out + tin : nullptr;
    T* youttile = y + tout;
    // carry mode: rows i0 .. i0 + OV - 1 of this band are the last OV staged rows of the previous one
    const int ov = (carry && (int)ti == prev_img && band == prev_band + 1) ? OV : 0;
    // a materialised block input (a_out) is stored by whoever STAGES a row: the row below this band too when the next band will take it over
    const int own_hi = (carry && t + 1 < t_end && band + 1 < nbands) ? i0 + nrows : o1;
    float4 cr[kCarryRegs];
    const int ncopy = ov * Wp * kSlabQuads;
    if (ov) {
      const float* src = lds + (size_t)(prev_nrows - OV) * Wp * kSlab;
#pragma unroll
      for (int u = 0; u < kCarryRegs; ++u)
        if (tid + u * kBlock < ncopy) cr[u] = ld4(src + (size_t)(tid + u * kBlock) * 4);
    }
    prev_img = (int)ti; prev_band = band; prev_nrows = nrows;
    __syncthreads();  // previous tile's readers are done (and the rows to carry over are in registers)
    if (ov) {
#pragma unroll
      for (int u = 0; u < kCarryRegs; ++u)
        if (tid + u * kBlock < ncopy) st4(lds + (size_t)(tid + u * kBlock) * 4, cr[u]);
    }
    // ---- stage: a_in rows i0 + ov .. i0+nrows-1, columns -1 .. W (zero outside the image) of nimg images.  Several
    // elements per thread and iteration so that their loads are in flight together (the staging phase is where this
    // kernel touches HBM).
    const int nstage = nimg * ((int)PI - ov * Wp) * kSlabQuads;
    const unsigned ovpix = (unsigned)(ov * Wp);
    constexpr int kFwdU = SKIP ? kFwdUSkip : (S == 2 ? kFwdUPlain2 : (CARRY ? kFwdUPlain - 1 : kFwdUPlain));  // (carry mode holds five more float4 across the barrier)
    for (int e = tid; e < nstage; e += kFwdU * kBlock) {
      float4 yv[kFwdU], sk[SKIP ? kFwdU : 1];
      unsigned off[kFwdU];
      bool in[kFwdU];
      int pxs[kFwdU], rows[kFwdU];
#pragma unroll
      for (int u = 0; u < kFwdU; ++u) {
        const int ee = e + u * kBlock;
        const unsigned pxa = ((unsigned)ee >> kQs) + ovpix;  // pixel slot in LDS over all images of the tile (carry mode: behind the rows taken over)
        const unsigned img = NI > 1 ? dPI.div(pxa) : 0u, px = NI > 1 ? pxa - __umul24(img, PI) : pxa;
        const unsigned prow = dWp.div(px);
        const int col = (int)(px - __umul24(prow, (unsigned)Wp)) - 1 + cx0, row = i0 + (int)prow;
        pxs[u] = (int)pxa;
        rows[u] = (row >= o0 && row < own_hi && col >= cx0 && col < cx0 + tw) ? 1 : 0;  // the one tile this input pixel belongs to
        in[u] = ee < nstage && row >= 0 && row < H && col >= 0 && col < W;
        off[u] = in[u] ? ((__umul24(__umul24(img, (unsigned)H) + (unsigned)row, (unsigned)W) + (unsigned)col) << cshift) + 4 * q : 0u;  // qq == q: kBlock is a multiple of 8
        yv[u] = in[u] ? Act<T>::ldnt(ytile + off[u]) : f4(0.f);
        if constexpr (SKIP) sk[u] = in[u] ? Act<T>::ldnt(sktile + off[u]) : f4(0.f);
      }
#pragma unroll
      for (int u = 0; u < kFwdU; ++u) {
        const int ee = e + u * kBlock;
        if (ee >= nstage) break;
        float4 a = f4(0.f);
        if (in[u]) {
          if constexpr (SKIP) a = bn.act(yv[u], sk[u]);
          else a = bn.act(yv[u]);
          if (S == 1 && a_out) a = Act<T>::round(a);  // a materialised block input is used as it is stored (residual, backward)
          if (S == 1 && a_out && rows[u]) Act<T>::st(aotile + off[u], a);
        }
        st4(lds + (size_t)pxs[u] * kSlab + 4 * q, a);
      }
    }
    __syncthreads();
    // ---- stencil
    const unsigned npix1 = (unsigned)((o1 - o0) * tw);  // output pixels per image
    const int npix = nimg * (int)npix1;
    const TileDiv dnp(npix1);
    for (int p = slot; p < npix; p += kPixSlots) {
      const unsigned img = NI > 1 ? dnp.div((unsigned)p) : 0u, pp = NI > 1 ? (unsigned)p - __umul24(img, npix1) : (unsigned)p;
      const unsigned prow = dtw.div(pp);
      const int ho = o0 + (int)prow, wl = (int)(pp - __umul24(prow, (unsigned)tw)), wo = cx0 + wl;
      const float* base = lds + ((__umul24(img, PI) + __umul24(prow * S, (unsigned)Wp) + (unsigned)(wl * S)) << kPs) + 4 * q;  // tap (0,0): row ho*S-1, col wo*S-1
      float4 acc = f4(0.f);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc = fma4(ld4(base + ((size_t)kh * Wp + kw) * kSlab), wr.tap(kh * 3 + kw), acc);
      acc = Act<T>::round(acc);  // statistics of what is stored
      Act<T>::st(youttile + ((__umul24(__umul24(img, (unsigned)Ho) + (unsigned)ho, (unsigned)Wo) + (unsigned)wo) << cshift) + 4 * q, acc);
      acc = sub4(acc, pv);
      s1.add(acc);
      s2.addmul(acc, acc);
    }
  }
  if (part) {
    const int stage = NI * ((R - 1) * S + 3) * (NCT > 1 ? TW + 2 : W + 2) * kSlab;
    slab_partials<SL>(s1, s2, q, C, slab * kSlab, part + (size_t)(blockIdx.x / nslabs) * 2 * C, lds + stage);
  }
}

// ---------------------------------------------------------------------------------------------
// data gradient (+ fused weight gradient)
// ---------------------------------------------------------------------------------------------
// LEAN: the block has no residual operands (a_in, skip_prev, skip_grad all null: the strided blocks and dw2_1).  Then the second phase
// reads one tensor only (yprev) and two pixels per iteration keep 8 KB per workgroup in flight: LEAN handles kLeanPix pixels per
// iteration with the registers the three absent operands would take.
#ifndef TTK_DW_BWD_LEAN_PIX
#define TTK_DW_BWD_LEAN_PIX 4   // stride 1 (168 registers: the cap)
#endif
#ifndef TTK_DW_BWD_LEAN_PIX2
#define TTK_DW_BWD_LEAN_PIX2 4  // stride 2 (6 measured slower: 276 vs 263 us on the 65 x 65 x 64 layer)
#endif
template <int S, typename T, typename TG, int SL, bool LEAN, bool CARRY>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(TTK_DW_WGS_PER_CU, TTK_DW_WGS_PER_CU)))  // <= 168 VGPRs: 3 workgroups per CU, as the LDS tile allows
dw_bwd_tiled_k(const TG* __restrict__ g_dw, const T* __restrict__ y_dw,
                                                          const float* __restrict__ bn_dw, const float* __restrict__ w,
                                                          const TG* __restrict__ skip_grad,
                                                          const T* __restrict__ yprev, float* __restrict__ bn_prev,
                                                          const T* __restrict__ skip_prev, const T* __restrict__ a_in,
                                                          TG* __restrict__ g_prev, float* __restrict__ part,
                                                          float* __restrict__ dwgrad, float* __restrict__ dw_partial, int B, int H, int W, int C, int Ho, int Wo,
                                                          int R, int nbands, int nslabs, int stage_floats, int NI_, int NCT_, int TW, int ring) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // dy[NI][stage_rows][Wo+2][SL] + reduction scratch
  const int NI = CARRY ? 1 : NI_, NCT = CARRY ? 1 : NCT_;  // (carry mode: one image per tile, full-width bands)
  constexpr int kSlab = SL, kSlabQuads = SL / 4, kPixSlots = kBlock / kSlabQuads, kQs = ilog2(kSlabQuads), kPs = ilog2(SL);
  const int tid = threadIdx.x, q = tid & (kSlabQuads - 1), slot = tid >> kQs;
  const int slab = blockIdx.x % nslabs, c0 = slab * kSlab + 4 * q;
  // the filter taps of this thread's channel quad live in LDS (wt[tap][32 channels], after the reduction scratch):
  // 36 fewer VGPRs, which pays for handling two pixels per iteration below
  float* wt = lds + stage_floats + 4 * 9 * kSlab;
  if (slot == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
      st4(wt + t * kSlab + 4 * q, make_float4(w[(size_t)(c0 + 0) * 9 + t], w[(size_t)(c0 + 1) * 9 + t], w[(size_t)(c0 + 2) * 9 + t], w[(size_t)(c0 + 3) * 9 + t]));
  }
  static_assert(SL == kCB, "a slab is one channel block of the activation layout");
  constexpr int cshift = 5;  // pixels of a channel block are 32 elements apart (ttk_common.h act_off)
  const BnApply4 bnp = BnApply4::load(bn_prev, C, c0);
  const BnGrad4 bg = BnGrad4::load(bn_dw, C, c0);
  D4 s1{0.0, 0.0, 0.0, 0.0}, s2{0.0, 0.0, 0.0, 0.0};
  float gmx = 0.f;  // max |g_prev|: the magnitude bound the previous block's fp16-split GEMMs scale by (ttk.h, TTK_AUX_GMAX)
  float4 wacc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wacc[t] = f4(0.f);
  const unsigned tiles = (unsigned)((B + NI - 1) / NI) * nbands * NCT;  // (< 2^31: checked by the host)
  // Tile order as in the forward kernel: round-robin, or - carry mode - a CONTIGUOUS run of tiles per workgroup (consecutive bands of an image), the
  // dy rows two neighbouring bands share going from one band's stage to the next through registers instead of being read (g AND y) and formed again.
  const unsigned wgs = gridDim.x / nslabs, wg = blockIdx.x / nslabs;
  const unsigned t_begin = CARRY ? (unsigned)((uint64_t)tiles * wg / wgs) : wg, t_end = CARRY ? (unsigned)((uint64_t)tiles * (wg + 1) / wgs) : tiles;
  const unsigned t_step = CARRY ? 1u : wgs;
  int prev_img = -1, prev_band = -2, prev_hi = -1;
  for (unsigned t = t_begin; t < t_end; t += t_step) {
    const unsigned tb = t / (unsigned)NCT, ti = tb / (unsigned)nbands;
    const int ct = (int)(t - tb * NCT), band = (int)(tb - ti * nbands), n0 = (int)ti * NI;  // NI > 1 implies one tile per image
    const int nimg = min(NI, B - n0);
    const int cx0 = ct * TW, tw = NCT > 1 ? min(TW, W - cx0) : W;  // input columns [cx0, cx0 + tw) (column tiles: stride 1)
    const int Wp = NCT > 1 ? tw + 2 : Wo + 2;                       // staged dy columns cx0-1 .. cx0+tw (or -1 .. Wo)
    const int r0 = band * R, r1 = min(r0 + R, H);
    // output rows ho with ho*S + kh - 1 in [r0, r1): ho in [ceil((r0-1)/S), floor(r1/S)], clipped
    const int ho_lo = max(0, (r0 - 1 + S - 1) / S * (r0 > 0 ? 1 : 0));
    const int ho_hi = min(Ho - 1, r1 / S);
    const int nrows = ho_hi - ho_lo + 1;
    const unsigned PI = (unsigned)(nrows * Wp);  // staged pixels per image
    const TileDiv dPI(PI), dWp((unsigned)Wp), dtw((unsigned)tw);
    // one 64-bit base per tile and tensor, 32-bit element offsets from 24-bit multiplies (see the forward kernel)
    // channel block `slab`, first pixel of image n0 (uniform: scalar registers)
    const size_t tdy = ((size_t)slab * B * Ho * Wo + (size_t)n0 * Ho * Wo) * kCB, tin = ((size_t)slab * B * H * W + (size_t)n0 * H * W) * kCB;
    const TG* gtile = g_dw + tdy;
    const T* ydtile = y_dw + tdy;
    const T* yptile = yprev + tin;
    const T* aitile = a_in ? a_in + tin : nullptr;
    const T* sktile = skip_prev ? skip_prev + tin : nullptr;
    const TG* sgtile = skip_grad ? skip_grad + tin : nullptr;
    TG* gptile = g_prev + tin;
    // carry mode: dy rows ho_lo .. ho_lo + ov - 1 of this band are the last ov staged rows of the previous one.  They STAY where they are: the
    // stage is a ring of `ring` rows (dy row ho of an image lives in ring row ho mod ring), a band stages only its new rows behind them.  (The
    // forward kernel hands its shared rows over through five float4 registers; here that broke the 168-register cap of three workgroups per
    // CU - 40 spilled registers - so the ring addressing pays three compare-and-subtracts per pixel instead.)
    const int ov = (CARRY && (int)ti == prev_img && band == prev_band + 1 && prev_hi >= ho_lo) ? prev_hi - ho_lo + 1 : 0;
    const int rbase = CARRY ? ho_lo % ring : 0;  // ring row of dy row ho_lo
    prev_img = (int)ti; prev_band = band; prev_hi = ho_hi;
    __syncthreads();  // the previous tile's readers are done
    // ---- stage dy rows ho_lo + ov..ho_hi, columns -1..Wo (zeros outside) of nimg images; kBwdU elements per thread and
    // iteration (2 kBwdU loads in flight)
    const unsigned ovpix = (unsigned)(ov * Wp);
    const int nstage = nimg * ((int)PI - (int)ovpix) * kSlabQuads;
    constexpr int kBwdU = TTK_DW_BWD_U;  // staged elements per thread and iteration: 2 * kBwdU loads in flight
    for (int e = tid; e < nstage; e += kBwdU * kBlock) {
      float4 gv[kBwdU], yv[kBwdU];
      bool in[kBwdU];
      unsigned pxs[kBwdU];
#pragma unroll
      for (int u = 0; u < kBwdU; ++u) {
        const int ee = e + u * kBlock;
        const unsigned pxa = ((unsigned)ee >> kQs) + ovpix;
        const unsigned img = NI > 1 ? dPI.div(pxa) : 0u, px = NI > 1 ? pxa - __umul24(img, PI) : pxa;
        const unsigned prow = dWp.div(px);
        const int col = (int)(px - __umul24(prow, (unsigned)Wp)) - 1 + cx0, row = ho_lo + (int)prow;
        if constexpr (CARRY) {
          unsigned rr = (unsigned)rbase + prow;
          rr -= rr >= (unsigned)ring ? (unsigned)ring : 0u;
          pxs[u] = __umul24(rr, (unsigned)Wp) + (px - __umul24(prow, (unsigned)Wp));
        } else {
          pxs[u] = pxa;
        }
        in[u] = ee < nstage && col >= 0 && col < Wo;
        const unsigned off = in[u] ? ((__umul24(__umul24(img, (unsigned)Ho) + (unsigned)row, (unsigned)Wo) + (unsigned)col) << cshift) + 4 * q : 0u;  // qq == q (see forward)
        gv[u] = in[u] ? Act<TG>::ldnt(gtile + off) : f4(0.f);
        yv[u] = in[u] ? Act<T>::ldnt(ydtile + off) : f4(0.f);
      }
#pragma unroll
      for (int u = 0; u < kBwdU; ++u) {
        const int ee = e + u * kBlock;
        if (ee >= nstage) break;
        st4(lds + (size_t)pxs[u] * kSlab + 4 * q, in[u] ? bg.dy(gv[u], yv[u]) : f4(0.f));
      }
    }
    __syncthreads();
    const unsigned npix1 = (unsigned)((r1 - r0) * tw);  // input pixels per image
    const int npix = nimg * (int)npix1;
    const TileDiv dnp(npix1);
    // NP pixels per thread and iteration: their (up to 3 NP) global loads are issued back to back before any pixel's LDS taps are read
    constexpr int NP = LEAN ? (S == 2 ? TTK_DW_BWD_LEAN_PIX2 : TTK_DW_BWD_LEAN_PIX) : 2;
    for (int p = slot; p < npix; p += NP * kPixSlots) {
      bool has[NP];
      unsigned imgs[NP], offs[NP];
      int his[NP], wis[NP];
      float4 yps[NP], raws[NP], sgs[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const int pj = p + j * kPixSlots;
        has[j] = pj < npix;
        const unsigned pq = has[j] ? (unsigned)pj : (unsigned)p;  // absent pixels repeat the first one's (valid) address
        imgs[j] = NI > 1 ? dnp.div(pq) : 0u;
        const unsigned pp = NI > 1 ? pq - __umul24(imgs[j], npix1) : pq;
        const unsigned pr = dtw.div(pp);
        his[j] = r0 + (int)pr;
        wis[j] = cx0 + (int)(pp - __umul24(pr, (unsigned)tw));
        offs[j] = ((__umul24(__umul24(imgs[j], (unsigned)H) + (unsigned)his[j], (unsigned)W) + (unsigned)wis[j]) << cshift) + 4 * q;
      }
#pragma unroll
      for (int j = 0; j < NP; ++j) yps[j] = Act<T>::ldnt(yptile + offs[j]);
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        raws[j] = f4(0.f);
        sgs[j] = f4(0.f);
        if (!LEAN) {
          if (a_in) raws[j] = Act<T>::ldnt(aitile + offs[j]);
          else if (skip_prev) raws[j] = Act<T>::ldnt(sktile + offs[j]);
          if (skip_grad) sgs[j] = Act<TG>::ldnt(sgtile + offs[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        if (j > 0 && !has[j]) break;
        const int hi = his[j], wi = wis[j];
        const float* dyimg = lds + (__umul24(imgs[j], PI) << kPs);
        const float4 yp = yps[j], raw = raws[j], sg = sgs[j];
        float4 a;
        if (!LEAN && a_in) a = raw;
        else a = (!LEAN && skip_prev) ? bnp.act(yp, raw) : bnp.act(yp);
        float4 G = f4(0.f);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int th = hi + 1 - kh;
          if (th < 0 || (S == 2 && (th & 1))) continue;
          const int ho = th / S;
          if (ho > Ho - 1) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int tw = wi + 1 - kw;  // -1 .. W
            if (S == 2 && (tw & 1)) continue;
            const int wo = (S == 1) ? tw : (tw >> 1);  // -1 or Wo hit the zero padding columns (S=1); always inside for S=2
            int lr = ho - ho_lo;  // staged row of this tap (carry mode: through the ring)
            if constexpr (CARRY) { lr += rbase; lr -= lr >= ring ? ring : 0; }
            const float4 dy = ld4(dyimg + ((__mul24(lr, Wp) + wo - cx0 + 1) << kPs) + 4 * q);
            G = fma4(dy, ld4(wt + (kh * 3 + kw) * kSlab + 4 * q), G);
            wacc[kh * 3 + kw] = fma4(dy, a, wacc[kh * 3 + kw]);
          }
        }
        if (!LEAN && skip_grad) G = add4(G, sg);
        const float4 gp = Act<TG>::round(mask4(G, a));  // sums and maximum of what is stored
        Act<TG>::st(gptile + offs[j], gp);
        gmx = fmaxf(fmaxf(gmx, fmaxf(fabsf(gp.x), fabsf(gp.y))), fmaxf(fabsf(gp.z), fabsf(gp.w)));
        s1.add(gp);
        s2.addmul(gp, sub4(yp, bnp.mean));
      }
    }
  }
  float* red = lds + stage_floats;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) gmx = fmaxf(gmx, __shfl_xor(gmx, off));
  if ((tid & 63) == 0) {  // most waves find the slot already at or above their maximum: one relaxed read instead of ~3000 atomics on one address
    unsigned* slot = reinterpret_cast<unsigned*>(bn_prev + (size_t)TTK_BN_AUX * C + TTK_AUX_GMAX);
    if (__float_as_uint(gmx) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(slot, __float_as_uint(gmx));
  }
  if (part) slab_partials<SL>(s1, s2, q, C, slab * kSlab, part + (size_t)(blockIdx.x / nslabs) * 2 * C, red);
  if (dwgrad) {
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float4 v = slab_wave_fold<SL>(wacc[t]);
      if (lane < kSlabQuads) st4(red + ((size_t)wv * 9 + t) * kSlab + 4 * q, v);
    }
    __syncthreads();
    for (int i = tid; i < 9 * kSlab; i += kBlock) {
      const int t = i / kSlab, c = i % kSlab;
      float a = 0.f;
      for (int wq = 0; wq < kBlock / kWave; ++wq) a += red[((size_t)wq * 9 + t) * kSlab + c];
      if (dw_partial) dw_partial[((size_t)(blockIdx.x / nslabs) * C + slab * kSlab + c) * 9 + t] = a;  // deterministic mode: folded by fold_partials_k
      else atomicAdd(dwgrad + (size_t)(slab * kSlab + c) * 9 + t, a);
    }
  }
}

__global__ void zero_fill_k(float* p, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

static bool dw_shape_ok2(int B, int H, int W, int C, int stride) {
  return B > 0 && H > 0 && W > 0 && W <= 256 && C >= 32 && C <= 1024 && (C & (C - 1)) == 0 && (stride == 1 || stride == 2);
}

}  // namespace ttk

using namespace ttk;

extern "C" {

int ttk_partial_rows_dwconv(int B, int H, int W, int C, int stride, int backward) {
  if (!dw_shape_ok2(B, H, W, C, stride)) return -1;
  return dw_tiling(B, H, W, C, stride, backward != 0).rows;
}

int ttk_dwconv3x3_fwd(const void* yprev, const float* bn_prev, const void* skip_prev, void* a_out, const float* w, void* y,
                      float* part, const float* pivot, int B, int H, int W, int C, int stride, int act_bf16, ttk_stream_t stream) {
  TTK_REQUIRE(yprev && bn_prev && w && y, "dwconv3x3_fwd: null pointer");
  TTK_REQUIRE(dw_shape_ok2(B, H, W, C, stride), "dwconv3x3_fwd: unsupported shape B=%d H=%d W=%d C=%d stride=%d (C: power of two in 32..1024, W <= 256)", B, H, W, C, stride);
  TTK_REQUIRE(!(a_out && stride != 1), "dwconv3x3_fwd: a_out requires stride 1");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const DwTiling t = dw_tiling(B, H, W, C, stride, false);
  TTK_REQUIRE((int64_t)((B + t.NI - 1) / t.NI) * t.nbands * t.NCT < ((int64_t)1 << 31), "dwconv3x3_fwd: too many tiles for 32-bit indexing");
  const size_t stage = (size_t)t.NI * t.stage_rows * ((t.NCT > 1 ? t.TW : W) + 2) * t.SL;
  const size_t sm = (stage + 16 * t.SL) * sizeof(float);  // + [4][2][SL] doubles of reduction scratch
#define TTK_DW_FWD_SL(S_, SK_, SL_, CY_)                                                                                                     \
  hipLaunchKernelGGL((dw_fwd_tiled_k<S_, ActT, SK_, SL_, CY_>), dim3(t.grid), dim3(kBlock), sm, (hipStream_t)stream, (const ActT*)yprev, bn_prev, \
                     (const ActT*)skip_prev, (ActT*)a_out, w, (ActT*)y, part, pivot, B, H, W, C, Ho, Wo, t.R, t.nbands, t.nslabs, t.NI, t.NCT, t.TW)
#define TTK_DW_FWD(S_, SK_) do { if (t.carry) TTK_DW_FWD_SL(S_, SK_, kCB, true); else TTK_DW_FWD_SL(S_, SK_, kCB, false); } while (0)
  TTK_ACT_DISPATCH(act_bf16, if (stride == 1) { if (skip_prev) TTK_DW_FWD(1, true); else TTK_DW_FWD(1, false); }
                             else { if (skip_prev) TTK_DW_FWD(2, true); else TTK_DW_FWD(2, false); });
#undef TTK_DW_FWD
#undef TTK_DW_FWD_SL
  TTK_LAUNCH_CHECK("dwconv3x3_fwd");
}

int ttk_dwconv3x3_bwd_data(const void* g_dw, const void* y_dw, const float* bn_dw, const float* w, const void* skip_grad,
                           const void* yprev, float* bn_prev, const void* skip_prev, const void* a_in, void* g_prev,
                           float* part, float* dw, int dw_accumulate, float* dw_partial, int B, int H, int W, int C, int stride,
                           int act_bf16, ttk_stream_t stream) {
  TTK_REQUIRE(g_dw && y_dw && bn_dw && w && yprev && bn_prev && g_prev, "dwconv3x3_bwd_data: null pointer");
  TTK_REQUIRE(dw_shape_ok2(B, H, W, C, stride), "dwconv3x3_bwd_data: unsupported shape");
  TTK_REQUIRE(!(skip_grad && stride != 1), "dwconv3x3_bwd_data: residual gradient requires stride 1");
  const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
  const DwTiling t = dw_tiling(B, H, W, C, stride, true);
  TTK_REQUIRE((int64_t)((B + t.NI - 1) / t.NI) * t.nbands * t.NCT < ((int64_t)1 << 31), "dwconv3x3_bwd_data: too many tiles for 32-bit indexing");
  const size_t stage = (size_t)t.NI * t.stage_rows * ((t.NCT > 1 ? t.TW : Wo) + 2) * t.SL;
  const size_t sm = (stage + 4 * 9 * t.SL + 9 * t.SL) * sizeof(float);  // stage + reduction scratch + filter taps
  hipStream_t st = (hipStream_t)stream;
  if (!dw) dw_partial = nullptr;
  TTK_REQUIRE(dw_accumulate != 2 || dw_partial, "dwconv3x3_bwd_data: dw_accumulate = 2 (rows folded by the caller) needs dw and dw_partial");
  if (dw && !dw_accumulate && !dw_partial) hipLaunchKernelGGL(zero_fill_k, dim3((9 * C + 255) / 256), dim3(256), 0, st, dw, (int64_t)9 * C);
#define TTK_DW_BWD(S_) TTK_DW_BWD_SL(S_, kCB)
#define TTK_DW_BWD_SL(S_, SL_) \
  TTK_DW_BWD_C(S_, SL_, true); else TTK_DW_BWD_C(S_, SL_, false)
#define TTK_DW_BWD_C(S_, SL_, LEAN_) do { if (t.carry) TTK_DW_BWD_L(S_, SL_, LEAN_, true); else TTK_DW_BWD_L(S_, SL_, LEAN_, false); } while (0)
#define TTK_DW_BWD_L(S_, SL_, LEAN_, CY_)                                                                                                 \
  hipLaunchKernelGGL((dw_bwd_tiled_k<S_, ActT, GradT, SL_, LEAN_, CY_>), dim3(t.grid), dim3(kBlock), sm, st, (const GradT*)g_dw, (const ActT*)y_dw, bn_dw, w, \
                     (const GradT*)skip_grad, (const ActT*)yprev, bn_prev, (const ActT*)skip_prev, (const ActT*)a_in, (GradT*)g_prev, part,  \
                     dw, dw_partial, B, H, W, C, Ho, Wo, t.R, t.nbands, t.nslabs, (int)stage, t.NI, t.NCT, t.TW, t.stage_rows)
  const bool lean = !a_in && !skip_prev && !skip_grad;
  TTK_ACT_DISPATCH(act_bf16, if (stride == 1) { if (lean) TTK_DW_BWD(1); } else { if (lean) TTK_DW_BWD(2); });
#undef TTK_DW_BWD_L
#undef TTK_DW_BWD_C
#undef TTK_DW_BWD
#undef TTK_DW_BWD_SL
  // dw_accumulate == 2: the rows stay unfolded - the caller folds them beside the BatchNorm-backward finalisation (ttk_bc_bn_bwd_finalize_fold)
  if (dw_partial && dw_accumulate != 2) launch_fold_partials(dw_partial, t.rows, (int64_t)9 * C, dw, dw_accumulate, st);
  TTK_LAUNCH_CHECK("dwconv3x3_bwd_data");
}

}  // extern "C"
