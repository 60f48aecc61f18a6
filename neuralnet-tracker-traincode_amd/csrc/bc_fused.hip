// Weight gradient AND data gradient of the early pointwise layers (32 -> 64, 64 -> 128, 128 -> 128) in ONE kernel on the bf16-COMPUTE path.
//
// As two kernels these layers - the largest activations of the network, HBM-bound - read g, y (twice) and ydw (twice) and write g_dw: seven
// tensor passes; here the workgroup stages a chunk of CP pixels of dy = bf16(ga*g + gb*y + c0), a = bf16(relu(scale*ydw + shift)) and the raw
// ydw ONCE in LDS (their own [pixel][channel] order, as bc_wgrad_k does) and forms both products from the tiles:
//   dW[co][ci]   += sum_m dy[m][co] a[m][ci]          transposed fragment reads (ds_read_b64_tr_b16), accumulators live until the end
//   g_dw[m][ci]   = (sum_co dy[m][co] W[co][ci]) * [a > 0]   weight image resident in LDS (MFMA "A" operand: rows = ci), dy fragments (columns
//                                                     = pixels) read row-wise from the same dy tile; per chunk the accumulators go through
//                                                     the wave's LDS store tile to global memory (bc_common.h store_block), mask operand and
//                                                     the second partial sum from the raw ydw tile in LDS
// four passes (three reads, one write).  A workgroup owns one slice of the pixels (tile = the whole Cout x Cin matrix); it stores its dW
// to partial[slice] (folded in a fixed order) and one row of BatchNorm-backward partial sums.
#include <type_traits>

#include "bc_common.h"

namespace ttk {
namespace bc {

template <int TN32, int TK32, int CP>
__global__ void __launch_bounds__(512) bc_bwd_fused_k(const bf16_t* __restrict__ G, const bf16_t* __restrict__ Y, const float* __restrict__ bn_pw,
                                                       const bf16_t* __restrict__ X, const float* __restrict__ bn_x, const uint4* __restrict__ Wd,
                                                       bf16_t* __restrict__ gdw, float* __restrict__ partial, float* __restrict__ part, int64_t M,
                                                       int64_t rows_per_slice) {
  constexpr int TN = 32 * TN32, TK = 32 * TK32, PN = wg_pitch(TN), PK = wg_pitch(TK);
  constexpr int ON = TN / 8, OK = TK / 8;
  constexpr int IN = CP * ON / 512, IK = CP * OK / 512;
  static_assert(IN >= 1 && IK >= 1 && CP * ON % 512 == 0 && CP * OK % 512 == 0, "chunk too small for 512 threads");
  constexpr int WN = TN32 >= 4 ? 4 : 2, WK = TK32 >= 2 ? 2 : 1, KS = 8 / (WN * WK);  // weight-gradient wave grid (bc_wgrad_k)
  constexpr int BN = TN32 / WN, BK = TK32 / WK;
  constexpr int kBuf = CP * (PN + 2 * PK);                      // dy | a | raw ydw
  constexpr int OC = TK < 64 ? 32 : 64, NCB = TK / OC, NPG = CP / 32, UNITS = NCB * NPG, BPC = OC / 32;  // data-gradient units: (channel block, pixel group)
  static_assert(UNITS <= 8, "one data-gradient unit per wave at most");
  constexpr int LPP = OC / 8, PPI = 64 / LPP, NI = 32 / PPI;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* wl_b = lds + 2 * kBuf;                          // dgrad weight image: [TN / 64][TK rows][8 chunks]
  const uint4* Wl = reinterpret_cast<const uint4*>(wl_b);
  float* cE = reinterpret_cast<float*>(wl_b + TK * TN * 2);      // scale | shift | mean of ydw's BatchNorm
  uint4* stg = reinterpret_cast<uint4*>(cE + 3 * TK);            // 8 waves x 4 OC chunks
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned slice = blockIdx.x;
  const int64_t m_begin = (int64_t)slice * rows_per_slice;
  const int64_t m_end = (m_begin + rows_per_slice < M) ? m_begin + rows_per_slice : M;
  const int nchunks = m_begin < m_end ? (int)((m_end - m_begin + CP - 1) / CP) : 0;
  for (int i = tid; i < TK * TN / 8; i += 512) reinterpret_cast<uint4*>(wl_b)[i] = Wd[i];
  fill_cE<kDgrad>(cE, nullptr, bn_x, TK, 0, TK, tid, 512);

  // ---- staging role (fixed chunk columns per thread: constants in registers)
  const int on = tid % ON, ok = tid % OK;
  f2 ga[4], gb[4], c0[4], sc[4], sh[4];
  {
    f2 gm[4], mu[4], be[4];
    ld8(bn_pw + TTK_BN_GA * TN + 8 * on, ga);
    ld8(bn_pw + TTK_BN_GB * TN + 8 * on, gb);
    ld8(bn_pw + TTK_BN_GMEAN * TN + 8 * on, gm);
    ld8(bn_pw + TTK_BN_MEAN * TN + 8 * on, mu);
#pragma unroll
    for (int k = 0; k < 4; ++k) c0[k] = -ga[k] * gm[k] - gb[k] * mu[k];
    ld8(bn_x + TTK_BN_SCALE * TK + 8 * ok, sc);
    ld8(bn_x + TTK_BN_MEAN * TK + 8 * ok, mu);
    ld8(bn_x + TTK_BN_BETA * TK + 8 * ok, be);
#pragma unroll
    for (int k = 0; k < 4; ++k) sh[k] = fma2(-sc[k], mu[k], be[k]);
  }
  constexpr int wo = TN < 64 ? TN : 64, wi = TK < 64 ? TK : 64;
  // per-thread bases at the slice's first pixel; a chunk's pixels are (uniform chunk offset) + (fixed pixel of this thread): 32-bit arithmetic, one
  // clamp against the slice's end (the loads of the last chunk stay inside the tensor; their values are zeroed by the `live` guard below)
  constexpr int PXN = 512 / ON, PXK = 512 / OK;  // pixels between a thread's items
  const int pn0 = tid / ON, pk0 = tid / OK, nrel = (int)(m_end - m_begin);
  const bf16_t* Gb = G + (size_t)((8 * on) / wo) * M * wo + ((8 * on) % wo) + (size_t)m_begin * wo;
  const bf16_t* Yb = Y + (size_t)((8 * on) / wo) * M * wo + ((8 * on) % wo) + (size_t)m_begin * wo;
  const bf16_t* Xb = X + (size_t)((8 * ok) / wi) * M * wi + ((8 * ok) % wi) + (size_t)m_begin * wi;
  // two register sets: chunk k travels in set k & 1, so chunks c + 2 AND c + 3 are in flight while chunk c is multiplied (one set = 32 - 48 KB per
  // CU in flight held the 128 -> 128 layer at 0.50 of 8 TB/s)
  u32x4 rg[2][IN], ry[2][IN], rx[2][IK];
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;
  auto load = [&](int c, auto rset) {
    constexpr int st = decltype(rset)::value;
#pragma unroll
    for (int i = 0; i < IN; ++i) {
      const unsigned rel = (unsigned)min(c * CP + i * PXN + pn0, nrel - 1);
      rg[st][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Gb + rel * (unsigned)wo));
      ry[st][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Yb + rel * (unsigned)wo));
    }
#pragma unroll
    for (int i = 0; i < IK; ++i) {
      const unsigned rel = (unsigned)min(c * CP + i * PXK + pk0, nrel - 1);
      rx[st][i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Xb + rel * (unsigned)wi));
    }
  };
  auto store_to = [&](int c, int slot, auto rset) {
    constexpr int st = decltype(rset)::value;
    unsigned char* buf = lds + slot * kBuf;
    const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < IN; ++i) {
      const int px = pn0 + i * PXN;
      const bool live = c * CP + px < nrel;  // pixels past the slice contribute nothing
      const uint4 d = dy_chunk(rg[st][i], ry[st][i], ga, gb, c0);
      st16(buf + px * PN + on * 16, live ? d : zero);
    }
#pragma unroll
    for (int i = 0; i < IK; ++i) {
      const int px = pk0 + i * PXK;
      const bool live = c * CP + px < nrel;
      const uint4 a = act_chunk(rx[st][i], sc, sh);
      st16(buf + CP * PN + px * PK + ok * 16, live ? a : zero);
      st16(buf + CP * (PN + PK) + px * PK + ok * 16, make_uint4(rx[st][i].x, rx[st][i].y, rx[st][i].z, rx[st][i].w));  // raw ydw: mask operand and second partial sum of the data gradient
    }
  };

  // ---- weight-gradient role (all waves)
  const int wsub = wave / (WN * WK), wq = wave % (WN * WK), wn = wq / WK, wk = wq % WK;
  const int grp = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3, h = grp >> 1;
  const int chan = 16 * (grp & 1) + 4 * p4;
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

  // ---- data-gradient role (waves 0 .. UNITS - 1): unit = (64-channel block ucb of the output, 32-pixel group upg of the chunk)
  const bool dg_wave = wave < UNITS;
  const int ucb = wave % NCB, upg = (wave / NCB) % NPG;
  const int r = lane & 31, hh = lane >> 5;
  const int wsw = (r >> 1) & 7;
  uint4* mystg = stg + wave * (4 * OC);
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;

  __syncthreads();  // weight image and constants are in LDS
  if (nchunks > 0) {
    const int clast = nchunks - 1;
    load(0, S0{});
    store_to(0, 0, S0{});
    load(min(1, clast), S1{});
    load(min(2, clast), S0{});
    __syncthreads();
    auto step = [&](int c, auto snext) {  // snext: the register set of chunk c + 1
      const unsigned char* buf = lds + (c & 1) * kBuf;
#pragma unroll
      for (int ks = 0; ks < CP / 16; ++ks) {
        if (ks % KS != wsub) continue;
        bf16x8 a[BN], b[BK];
#pragma unroll
        for (int i = 0; i < BN; ++i) a[i] = tr_frag(buf, aoff[i] + ks * 16 * PN, PN);
#pragma unroll
        for (int j = 0; j < BK; ++j) b[j] = tr_frag(buf, boff[j] + ks * 16 * PK, PK);
#pragma unroll
        for (int i = 0; i < BN; ++i)
#pragma unroll
          for (int j = 0; j < BK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
      }
      if (dg_wave) {
        f32x16 gd[BPC];
#pragma unroll
        for (int bl = 0; bl < BPC; ++bl)
#pragma unroll
          for (int e = 0; e < 16; ++e) gd[bl][e] = 0.f;
        const unsigned char* dyrow = buf + (32 * upg + r) * PN + 16 * hh;  // this lane's pixel, its half of a k16 step
#pragma unroll
        for (int kb = 0; kb < TN / 64; ++kb)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const uint4 dv = *reinterpret_cast<const uint4*>(dyrow + (kb * 64 + 16 * s) * 2);
#pragma unroll
            for (int bl = 0; bl < BPC; ++bl) {
              const uint4 wv = Wl[(kb * TK + ucb * OC + 32 * bl + r) * 8 + ((2 * s + hh) ^ wsw)];
              gd[bl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, dv), gd[bl], 0, 0, 0);
            }
          }
        // mask operand (raw ydw) of the pixels / chunk column this lane will store: from the LDS tile
        uint4 mk[NI];
        const unsigned char* xraw = buf + CP * (PN + PK) + (32 * upg) * PK + (ucb * OC + 8 * (lane % LPP)) * 2;
#pragma unroll
        for (int i = 0; i < NI; ++i) mk[i] = *reinterpret_cast<const uint4*>(xraw + (PPI * i + lane / LPP) * PK);
        const int64_t g0 = m_begin + (int64_t)c * CP + 32 * upg;
        if (g0 < m_end) store_block<kDgrad, OC>(gd, mystg, gdw + ((size_t)ucb * M + g0) * OC, mk, cE + ucb * OC, TK, g0, m_end, s1, s2);
      }
      store_to(min(c + 1, clast), (c + 1) & 1, snext);  // (the other buffer: last read in step c - 1, every wave has passed the barrier since)
      load(min(c + 3, clast), snext);
      __syncthreads();
    };
    for (int c = 0; c < nchunks; c += 2) {
      step(c, S1{});
      if (c + 1 < nchunks) step(c + 1, S0{});
    }
  }
  // ---- partial sums of the data gradient: lanes of one chunk column, then the waves of a channel block (fixed order)
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int off = LPP; off < 64; off <<= 1) {
      s1[j] += __shfl_xor(s1[j], off);
      s2[j] += __shfl_xor(s2[j], off);
    }
  float* red = reinterpret_cast<float*>(lds);  // the chunk buffers are free (barrier above)
  if (dg_wave && lane < LPP) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[(wave * 2 + 0) * OC + 8 * lane + j] = s1[j];
      red[(wave * 2 + 1) * OC + 8 * lane + j] = s2[j];
    }
  }
  __syncthreads();
  if (part)
    for (int i = tid; i < 2 * TK; i += 512) {
      const int which = i / TK, cc = i - which * TK, cb = cc / OC, co = cc % OC;
      float a = 0.f;
#pragma unroll
      for (int u = 0; u < UNITS; ++u)
        if (u % NCB == cb) a += red[(u * 2 + which) * OC + co];
      part[(size_t)slice * 2 * TK + i] = a;
    }
  __syncthreads();
  // ---- the weight-gradient tile of this slice
  float* dst = partial + (size_t)slice * TN * TK;
  if constexpr (KS > 1) {
    float* redw = reinterpret_cast<float*>(lds);  // [8 waves][BN * BK blocks][16][64]
#pragma unroll
    for (int i = 0; i < BN; ++i)
#pragma unroll
      for (int j = 0; j < BK; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) redw[((wave * BN * BK + i * BK + j) * 16 + e) * 64 + lane] = acc[i][j][e];
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
            for (int s = 1; s < KS; ++s) a += redw[(((wave + s * WN * WK) * BN * BK + i * BK + j) * 16 + e) * 64 + lane];
            acc[i][j][e] = a;
          }
    }
  }
  if (wsub == 0) {
#pragma unroll
    for (int i = 0; i < BN; ++i)
#pragma unroll
      for (int j = 0; j < BK; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (wn * BN + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          const int col = (wk * BK + j) * 32 + r;
          dst[(size_t)row * TK + col] = acc[i][j][e];
        }
  }
}

struct FPlan { int CP; int64_t slices, rows; };
static bool fused_plan(int64_t M, int Cin, int Cout, FPlan& p) {
  if (Cout == 64 && Cin == 32) p.CP = 128;
  else if (Cout == 128 && Cin == 64) p.CP = 64;
  else if (Cout == 128 && Cin == 128) p.CP = 32;
  else return false;
  if (M < 1) return false;
  int64_t slices = 256;
  const int64_t max_slices = ceil_div(M, 2 * p.CP);
  if (slices > max_slices) slices = max_slices;
  p.rows = ceil_div(ceil_div(M, slices), p.CP) * p.CP;
  p.slices = ceil_div(M, p.rows);
  return true;
}
template <int TN32, int TK32, int CP>
static size_t fused_lds() {
  constexpr int TN = 32 * TN32, TK = 32 * TK32, OC = TK < 64 ? 32 : 64;
  constexpr size_t bufs = (size_t)2 * CP * (wg_pitch(TN) + 2 * wg_pitch(TK)), tail = (size_t)TK * TN * 2 + 3 * TK * 4 + 8 * 4 * OC * 16;
  constexpr int WN = TN32 >= 4 ? 4 : 2, WK = TK32 >= 2 ? 2 : 1, KS = 8 / (WN * WK);
  constexpr size_t redw = KS > 1 ? (size_t)8 * (TN32 / WN) * (TK32 / WK) * 16 * 64 * 4 : 0;
  return (bufs > redw ? bufs : redw) + tail;
}

template <int TN32, int TK32, int CP>
static void launch_fused(const FPlan& p, hipStream_t st, const void* g, const void* y, const float* bn_pw, const void* ydw, const float* bn_dw, const uint4* img,
                         void* g_dw, float* scratch, float* part, int64_t M) {
  constexpr auto kern = bc_bwd_fused_k<TN32, TK32, CP>;
  (void)allow_big_lds<kern>();
  const size_t lds_bytes = fused_lds<TN32, TK32, CP>();
  hipLaunchKernelGGL(kern, dim3((unsigned)p.slices), dim3(512), lds_bytes, st, (const bf16_t*)g, (const bf16_t*)y, bn_pw, (const bf16_t*)ydw,
                     bn_dw, img, (bf16_t*)g_dw, scratch, part, M, p.rows);
}

}  // namespace bc
}  // namespace ttk

using namespace ttk;
using namespace ttk::bc;

extern "C" {

/* rows of BatchNorm-backward partial sums (= pixel slices) the fused kernel writes; 0: this shape has no fused form */
int ttk_bc_pw_bwd_fused_rows(int64_t M, int Cin, int Cout) {
  FPlan p;
  return fused_plan(M, Cin, Cout, p) ? (int)p.slices : 0;
}
size_t ttk_bc_pw_bwd_fused_scratch_bytes(int64_t M, int Cin, int Cout) {
  FPlan p;
  return fused_plan(M, Cin, Cout, p) ? (size_t)p.slices * Cin * Cout * sizeof(float) : 0;
}

int ttk_bc_pw_bwd_fused(const void* g, const void* y, const float* bn_pw, const void* wprep, const void* ydw, const float* bn_dw, void* g_dw, float* dw,
                        float* scratch, float* part, int64_t M, int Cin, int Cout, ttk_stream_t stream) {
  TTK_REQUIRE(g && y && bn_pw && wprep && ydw && bn_dw && g_dw && scratch && part, "bc_pw_bwd_fused: null pointer");
  FPlan p;
  TTK_REQUIRE(fused_plan(M, Cin, Cout, p), "bc_pw_bwd_fused: no fused form for %d -> %d", Cin, Cout);
  hipStream_t st = (hipStream_t)stream;
  const uint4* img = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(wprep) + (size_t)Cin * Cout * 2);  // image 1: rows = Cin, k = Cout
  if (Cout == 64) launch_fused<2, 1, 128>(p, st, g, y, bn_pw, ydw, bn_dw, img, g_dw, scratch, part, M);
  else if (Cin == 64) launch_fused<4, 2, 64>(p, st, g, y, bn_pw, ydw, bn_dw, img, g_dw, scratch, part, M);
  else launch_fused<4, 4, 32>(p, st, g, y, bn_pw, ydw, bn_dw, img, g_dw, scratch, part, M);
  const int64_t n = (int64_t)Cin * Cout;
  // (dw == NULL: the caller folds the ttk_bc_pw_bwd_fused_rows tiles of scratch itself - ttk_bc_bn_bwd_finalize_fold)
  if (dw && !launch_fold_rows_fast(scratch, (int)p.slices, n, dw, 1, st)) launch_fold_partials(scratch, (int)p.slices, n, dw, 1, st);
  TTK_LAUNCH_CHECK("bc_pw_bwd_fused");
}

}  // extern "C"
