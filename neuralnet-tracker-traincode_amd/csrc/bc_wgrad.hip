// Weight gradient of the pointwise convolutions on the bf16-COMPUTE path (bc_common.h):
//     dW[co][ci] += sum_m dy[m][co] * a[m][ci],   dy = bf16(ga*g + gb*y + c0),  a = bf16(relu(scale*ydw + shift))   formed on load
// The contraction runs over the PIXELS, so an MFMA fragment needs 8 consecutive pixels of one channel while memory has the channels of a
// pixel contiguous: the workgroup stages chunks of CP pixels of both operands in LDS in their own [pixel][channel] order (16-byte stores,
// conflict-free) and the waves read their fragments TRANSPOSED (ds_read_b64_tr_b16: a 4 pixel x 16 channel block per 16-lane group;
// rows padded so that the four pixel rows of a block fall into different 64-byte bank windows).  Every thread stages a FIXED 16-byte chunk
// (8 channels) of each operand, so its BatchNorm constants live in registers; the loads of chunk i + 1 are in flight while chunk i is
// multiplied (double-buffered LDS, one barrier per chunk).  A workgroup owns one TN x TK tile of dW and one slice of the pixels; it stores
// its tile to partial[slice][Cout][Cin] and bc_wgrad_fold_k adds the slices to dW in a fixed order (bitwise reproducible; 256 workgroups
// of float atomics on one tile would cost more than the plain stores and the fold).
#include "bc_common.h"

// Timing-only bits (experiment builds: tools/exp/build_variants.sh ... "-DTTK_BC_WDBG=<bits>"; wrong results):
//   1 no MFMAs   2 no global loads   4 no staging (BatchNorm maps + LDS stores)   8 no fragment reads   16 no tile stores
#ifndef TTK_BC_WDBG
#define TTK_BC_WDBG 0
#endif

namespace ttk {
namespace bc {

// TN32 x TK32 blocks of 32 x 32 per tile; CP pixels per chunk; the 8 waves form a WN x WK grid over the blocks, KS of them share a block
// and split the k16 steps of a chunk between them (tiles of fewer than 8 blocks)
template <int TN32, int TK32, int CP, int WN, int WK, int KS>
__global__ void __launch_bounds__(512) bc_wgrad_k(const bf16_t* __restrict__ G, const bf16_t* __restrict__ Y, const float* __restrict__ bn_pw,
                                                   const bf16_t* __restrict__ X, const float* __restrict__ bn_x, float* __restrict__ partial, int64_t M,
                                                   int Cin, int Cout, int64_t rows_per_slice, int tiles) {
  static_assert(WN * WK * KS == 8, "eight waves");
  constexpr int TN = 32 * TN32, TK = 32 * TK32, PN = wg_pitch(TN), PK = wg_pitch(TK);
  constexpr int ON = TN / 8, OK = TK / 8;                   // 16-byte chunks per pixel of each operand
  constexpr int IN = CP * ON / 512, IK = CP * OK / 512;     // staged items per thread and chunk
  static_assert(IN >= 1 && IK >= 1 && CP * ON % 512 == 0 && CP * OK % 512 == 0, "chunk too small for 512 threads");
  constexpr int BN = TN32 / WN, BK = TK32 / WK;             // blocks per wave
  constexpr int kBuf = CP * (PN + PK);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // [2][dy plane | a plane]
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order: every XCD gets whole slices (the tiles of a slice read the same pixels)
  const unsigned NG = gridDim.x, Lid = blockIdx.x;
  const unsigned xq = NG / 8, xr = NG % 8, xcd = Lid % 8;
  const unsigned logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + Lid / 8;
  const unsigned tile = logical % tiles, slice = logical / tiles;
  const int tiles_k = Cin / TK;
  const int n0 = (tile / tiles_k) * TN, k0 = (tile % tiles_k) * TK;
  const int64_t m_begin = (int64_t)slice * rows_per_slice;
  const int64_t m_end = (m_begin + rows_per_slice < M) ? m_begin + rows_per_slice : M;
  const int nchunks = m_begin < m_end ? (int)((m_end - m_begin + CP - 1) / CP) : 0;

  // ---- staging role: chunk `on` of the dy operand, chunk `ok` of the a operand (fixed per thread)
  const int on = tid % ON, ok = tid % OK;
  const int cn = n0 + 8 * on, ck = k0 + 8 * ok;
  f2 ga[4], gb[4], c0[4], sc[4], sh[4];
  {
    f2 gm[4], mu[4], be[4];
    ld8(bn_pw + TTK_BN_GA * Cout + cn, ga);
    ld8(bn_pw + TTK_BN_GB * Cout + cn, gb);
    ld8(bn_pw + TTK_BN_GMEAN * Cout + cn, gm);
    ld8(bn_pw + TTK_BN_MEAN * Cout + cn, mu);
#pragma unroll
    for (int k = 0; k < 4; ++k) c0[k] = -ga[k] * gm[k] - gb[k] * mu[k];
    ld8(bn_x + TTK_BN_SCALE * Cin + ck, sc);
    ld8(bn_x + TTK_BN_MEAN * Cin + ck, mu);
    ld8(bn_x + TTK_BN_BETA * Cin + ck, be);
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[k] = fma2(-sc[k], mu[k], be[k]);
  }
  // per-thread bases at the slice's first pixel; a chunk's pixels are (uniform chunk offset) + (fixed pixel of this thread): 32-bit arithmetic, one
  // clamp against the slice's end (the loads of the last chunk stay inside the tensor; their values are zeroed by the `live` guard below)
  constexpr int wo = 64, wi = TK < 64 ? TK : 64;  // channels per block of the two tensors (TN >= 64)
  constexpr int PXN = 512 / ON, PXK = 512 / OK;
  const int pn0 = tid / ON, pk0 = tid / OK, nrel = (int)(m_end - m_begin);
  const size_t goff = (size_t)(cn / wo) * M * wo + (cn % wo) + (size_t)m_begin * wo;
  const bf16_t* Gb = G + goff;
  const bf16_t* Yb = Y + goff;
  const bf16_t* Xb = X + (size_t)(ck / wi) * M * wi + (ck % wi) + (size_t)m_begin * wi;
  u32x4 rg[IN], ry[IN], rx[IK];
  auto load = [&](int c) {
    if (TTK_BC_WDBG & 2) {
#pragma unroll
      for (int i = 0; i < IN; ++i) rg[i] = ry[i] = u32x4{(unsigned)c, (unsigned)i, (unsigned)tid, 1u};
#pragma unroll
      for (int i = 0; i < IK; ++i) rx[i] = u32x4{(unsigned)c, (unsigned)i, (unsigned)tid, 2u};
      return;
    }
#pragma unroll
    for (int i = 0; i < IN; ++i) {
      const unsigned rel = (unsigned)min(c * CP + i * PXN + pn0, nrel - 1);
      rg[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Gb + rel * (unsigned)wo));
      ry[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Yb + rel * (unsigned)wo));
    }
#pragma unroll
    for (int i = 0; i < IK; ++i) {
      const unsigned rel = (unsigned)min(c * CP + i * PXK + pk0, nrel - 1);
      rx[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Xb + rel * (unsigned)wi));
    }
  };
  auto store_to = [&](int c, int slot) {
    if (TTK_BC_WDBG & 4) {
      asm volatile("" ::"v"(rg[0].x), "v"(ry[0].x), "v"(rx[0].x));
      return;
    }
    unsigned char* buf = lds + slot * kBuf;
    const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < IN; ++i) {
      const int px = pn0 + i * PXN;
      const bool live = c * CP + px < nrel;  // pixels past the slice contribute nothing
      const uint4 d = dy_chunk(rg[i], ry[i], ga, gb, c0);
      st16(buf + px * PN + on * 16, live ? d : zero);
    }
#pragma unroll
    for (int i = 0; i < IK; ++i) {
      const int px = pk0 + i * PXK;
      const bool live = c * CP + px < nrel;
      const uint4 a = act_chunk(rx[i], sc, sh);
      st16(buf + CP * PN + px * PK + ok * 16, live ? a : zero);
    }
  };
  auto store = [&](int c) { store_to(c, c & 1); };

  // ---- multiply role
  const int wsub = wave / (WN * WK), wq = wave % (WN * WK), wn = wq / WK, wk = wq % WK;
  const int grp = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3, h = grp >> 1;
  const int chan = 16 * (grp & 1) + 4 * p4;  // first of the lane's 4 channels inside a 32-channel block
  int aoff[BN], boff[BK];
#pragma unroll
  for (int i = 0; i < BN; ++i) aoff[i] = (8 * h + q) * PN + ((wn * BN + i) * 32 + chan) * 2;
#pragma unroll
  for (int j = 0; j < BK; ++j) boff[j] = CP * PN + (8 * h + q) * PK + ((wk * BK + j) * 32 + chan) * 2;
  f32x16 acc[BN][BK];
#pragma unroll
  for (int i = 0; i < BN; ++i)
#pragma unroll
    for (int j = 0; j < BK; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nchunks > 0) {
    // every load / store of the pipeline is unconditional (chunk indices clamped to the last one; the surplus copies are never read): with
    // branches around them hipcc's counted vmcnt waits degrade to waiting for the loads just issued
    const int clast = nchunks - 1;
    load(0);
    store(0);
    load(min(1, clast));
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      const unsigned char* buf = lds + (c & 1) * kBuf;
#pragma unroll
      for (int ks = 0; ks < CP / 16; ++ks) {
        if (ks % KS != wsub) continue;
        bf16x8 a[BN], b[BK];
#pragma unroll
        for (int i = 0; i < BN; ++i) a[i] = (TTK_BC_WDBG & 8) ? __builtin_bit_cast(bf16x8, make_uint4(c, ks, i, lane)) : tr_frag(buf, aoff[i] + ks * 16 * PN, PN);
#pragma unroll
        for (int j = 0; j < BK; ++j) b[j] = (TTK_BC_WDBG & 8) ? __builtin_bit_cast(bf16x8, make_uint4(c, ks, j, lane)) : tr_frag(buf, boff[j] + ks * 16 * PK, PK);
#pragma unroll
        for (int i = 0; i < BN; ++i)
#pragma unroll
          for (int j = 0; j < BK; ++j) {
            if (TTK_BC_WDBG & 1) { acc[i][j][0] += __uint_as_float(__builtin_bit_cast(uint4, a[i]).x ^ __builtin_bit_cast(uint4, b[j]).y); continue; }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
          }
      }
      store_to(min(c + 1, clast), (c + 1) & 1);  // (the other buffer: last read in step c - 1, every wave has passed the barrier since)
      load(min(c + 2, clast));
      __syncthreads();
    }
  }
  // ---- the tile of this slice
  float* dst = partial + (size_t)slice * Cout * Cin;
  const int r = lane & 31, hh = lane >> 5;
  if constexpr (KS > 1) {  // the KS waves of a block add their parts through LDS (fixed order)
    float* red = reinterpret_cast<float*>(lds);  // [8 waves][BN * BK blocks][16][64]
    __syncthreads();
#pragma unroll
    for (int i = 0; i < BN; ++i)
#pragma unroll
      for (int j = 0; j < BK; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[((wave * BN * BK + i * BK + j) * 16 + e) * 64 + lane] = acc[i][j][e];
    __syncthreads();
    if (wsub == 0) {
#pragma unroll
      for (int i = 0; i < BN; ++i)
#pragma unroll
        for (int j = 0; j < BK; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            float a = acc[i][j][e];
#pragma unroll
            for (int s = 1; s < KS; ++s) a += red[(((wave + s * WN * WK) * BN * BK + i * BK + j) * 16 + e) * 64 + lane];
            acc[i][j][e] = a;
          }
    }
  }
  if (wsub == 0 && !((TTK_BC_WDBG & 16) && acc[0][0][0] != 12345.f)) {
#pragma unroll
    for (int i = 0; i < BN; ++i)
#pragma unroll
      for (int j = 0; j < BK; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = n0 + (wn * BN + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          const int col = k0 + (wk * BK + j) * 32 + r;
#if defined(TTK_BC_WGRAD_ATOMIC)  // (experiment builds: float atomics straight into dW instead of slice tiles + fold; `partial` = dW)
          unsafeAtomicAdd(partial + (size_t)row * Cin + col, acc[i][j][e]);
#else
          dst[(size_t)row * Cin + col] = acc[i][j][e];
#endif
        }
  }
}

// dW[i] += partial[0][i] + partial[1][i] + ... (fixed order); 16 B per lane, eight loads in flight
__global__ void __launch_bounds__(256) bc_wgrad_fold_k(const float* __restrict__ partial, float* __restrict__ dW, int64_t n, int slices) {
  fold_rows_wide_body(partial, slices, n, dW, blockIdx.x, 256);
}

struct WPlan { int TN, TK, CP, tiles; int64_t slices, rows; };
static bool wgrad_plan(int64_t M, int Cin, int Cout, WPlan& p) {
  auto p2 = [](int v) { return v >= 32 && v <= 1024 && (v & (v - 1)) == 0; };
  if (!p2(Cin) || !p2(Cout) || M < 1) return false;
  p.TN = Cout < 256 ? Cout : 256;
  p.TK = Cin < 256 ? Cin : 256;
  if (p.TN == 64 && p.TK == 32) p.CP = 128;
  else if (p.TN == 128 && p.TK <= 128) p.CP = 64;
  else if (p.TN == 256) p.CP = 32;
  else return false;
  if (p.TN == 128 && !(p.TK == 64 || p.TK == 128)) return false;
  if (p.TN == 256 && !(p.TK == 128 || p.TK == 256)) return false;
  p.tiles = (Cout / p.TN) * (Cin / p.TK);
  int64_t slices = 256 / p.tiles;
  if (slices < 1) slices = 1;
  const int64_t max_slices = ceil_div(M, 2 * p.CP);
  if (slices > max_slices) slices = max_slices;
  p.rows = ceil_div(ceil_div(M, slices), p.CP) * p.CP;
  p.slices = ceil_div(M, p.rows);
  return true;
}

}  // namespace bc
}  // namespace ttk

using namespace ttk;
using namespace ttk::bc;
#if defined(TTK_BC_WGRAD_ATOMIC)
#define TTK_BC_WGRAD_DST dw
#else
#define TTK_BC_WGRAD_DST scratch
#endif

extern "C" {

size_t ttk_bc_pw_wgrad_scratch_bytes(int64_t M, int Cin, int Cout) {
  WPlan p;
  if (!wgrad_plan(M, Cin, Cout, p)) return 0;
  return (size_t)p.slices * Cin * Cout * sizeof(float);
}

int ttk_bc_pw_wgrad_slices(int64_t M, int Cin, int Cout) {
  WPlan p;
  return wgrad_plan(M, Cin, Cout, p) ? (int)p.slices : 0;
}

int ttk_bc_pw_bwd_weight(const void* g, const void* y, const float* bn_pw, const void* ydw, const float* bn_dw, float* dw, float* scratch, int64_t M,
                         int Cin, int Cout, ttk_stream_t stream) {
  TTK_REQUIRE(g && y && bn_pw && ydw && bn_dw && scratch, "bc_pw_bwd_weight: null pointer");
  WPlan p;
  TTK_REQUIRE(wgrad_plan(M, Cin, Cout, p), "bc_pw_bwd_weight: unsupported shape M=%lld %d -> %d", (long long)M, Cin, Cout);
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)(p.tiles * p.slices);
#define TTK_BC_WG(TN32_, TK32_, CP_, WN_, WK_, KS_)                                                                                            \
  do {                                                                                                                                          \
    constexpr size_t sm = (size_t)2 * CP_ * (wg_pitch(32 * TN32_) + wg_pitch(32 * TK32_));                                                      \
    constexpr size_t red = KS_ > 1 ? (size_t)8 * (TN32_ / WN_) * (TK32_ / WK_) * 16 * 64 * 4 : 0;                                               \
    (void)allow_big_lds<bc_wgrad_k<TN32_, TK32_, CP_, WN_, WK_, KS_>>();                                                                              \
    hipLaunchKernelGGL((bc_wgrad_k<TN32_, TK32_, CP_, WN_, WK_, KS_>), dim3(grid), dim3(512), sm > red ? sm : red, st, (const bf16_t*)g, (const bf16_t*)y, bn_pw, \
                       (const bf16_t*)ydw, bn_dw, TTK_BC_WGRAD_DST, M, Cin, Cout, p.rows, p.tiles);                                                      \
  } while (0)
  if (p.TN == 64) TTK_BC_WG(2, 1, 128, 2, 1, 4);
  else if (p.TN == 128 && p.TK == 64) TTK_BC_WG(4, 2, 64, 4, 2, 1);
  else if (p.TN == 128) TTK_BC_WG(4, 4, 64, 4, 2, 1);
  else if (p.TK == 128) TTK_BC_WG(8, 4, 32, 4, 2, 1);
  else TTK_BC_WG(8, 8, 32, 4, 2, 1);
#undef TTK_BC_WG
#if !defined(TTK_BC_WGRAD_ATOMIC)
  if (!dw) {  // the caller folds the ttk_bc_pw_wgrad_slices tiles of scratch itself (ttk_bc_bn_bwd_finalize_fold)
    TTK_LAUNCH_CHECK("bc_pw_bwd_weight");
  }
  const int64_t n = (int64_t)Cin * Cout;
  if (!launch_fold_rows_fast(scratch, (int)p.slices, n, dw, 1, st))
    hipLaunchKernelGGL(bc_wgrad_fold_k, dim3((unsigned)ceil_div(n, 1024)), dim3(256), 0, st, scratch, dw, n, (int)p.slices);
#endif
  TTK_LAUNCH_CHECK("bc_pw_bwd_weight");
}

}  // extern "C"
