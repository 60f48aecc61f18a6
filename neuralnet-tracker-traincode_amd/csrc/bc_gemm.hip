// Pointwise 1x1 convolutions of the bf16-COMPUTE path (bc_common.h): forward and data gradient as ONE bf16 MFMA product.
//
//   forward  y[m][n]    = sum_k a[m][k] W[n][k]            a  = bf16(relu(scale*ydw + shift))            on load
//   dgrad    g_dw[m][n] = sum_k dy[m][k] W[k][n] * mask    dy = bf16(ga*g + gb*y + c0), mask = [a_dw > 0]  on load / in the epilogue
// (k = contraction channels: Cin forward, Cout data gradient; n = output channels).  Both are the same kernel: the weight image
// (ttk_bc_prepare_weights) is the MFMA "A" operand - rows = OUTPUT CHANNELS - and the activations are the "B" operand - columns =
// PIXELS - so that
//   * a lane's B fragment (8 consecutive channels of one pixel) is 16 contiguous bytes of the [C/64][M][64] tensor: the waves load their
//     fragments STRAIGHT from global memory into registers, transform them there and never stage activations in LDS;
//   * an accumulator lane holds 4 consecutive channels of one pixel: the epilogue packs them, transposes 32 pixels x 64 channels
//     through a wave-private 4 KB LDS tile and every lane stores 16 contiguous bytes - 1 KB per wave instruction, whole 128-byte lines.
// BatchNorm partial sums come from the values as stored (read back from that tile).
//
// Two structures, chosen per shape (all of them HBM- or CU-ingest-bound with one product):
//   bc_gemm_e_k  N <= 128 (K <= 256): the whole weight image is resident in LDS, the eight waves of a workgroup are INDEPENDENT
//                streams over 32-pixel groups (no barrier after the prologue), the next group's loads are in flight while one is computed;
//   bc_gemm_l_k  N a multiple of 256: tiles of <= 256 pixels x 256 channels, the weight image streamed per k64 step through a
//                two-slot LDS ring (plain loads one step ahead + ds_write), one barrier per step.
#include "bc_common.h"

// Timing-only bits of bc_gemm_l_k's main loop (experiment builds: tools/exp/build_variants.sh ... "-DTTK_BC_GDBG=<bits>"; wrong results):
//   1 no MFMAs   2 no global loads   4 no operand staging (BatchNorm map + LDS stores)   8 no fragment reads   16 no epilogue (mask loads, stores, sums)
#ifndef TTK_BC_GDBG
#define TTK_BC_GDBG 0
#endif

namespace ttk {
namespace bc {


// ---------------------------------------------------------------------------------------------
// weight images: [img 0: forward rows = Cout, k = Cin | img 1: data gradient rows = Cin, k = Cout], bf16, swizzled chunks
// ---------------------------------------------------------------------------------------------
constexpr int kMaxPrep = 16;
struct PrepTable {
  const float* w[kMaxPrep];
  uint4* out[kMaxPrep];
  int cin[kMaxPrep], cout[kMaxPrep];
  long long start[kMaxPrep + 1];  // first chunk of layer i in the flat index space
  int n;
};
__global__ void __launch_bounds__(256) bc_prepare_k(PrepTable t) {
  const long long total = t.start[t.n];
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    int l = 0;
    while (l + 1 < t.n && idx >= t.start[l + 1]) ++l;
    const int cin = t.cin[l], cout = t.cout[l];
    const long long local = idx - t.start[l], per = (long long)cin * cout / 8;
    const int img = (int)(local / per);
    const long long cl = local - (long long)img * per;
    const int K = img == 0 ? cin : cout, N = img == 0 ? cout : cin;
    const int cpr = w_cpr(K);
    const long long row = cl / cpr;
    const int cs = (int)(cl - row * cpr);
    const int kb = (int)(row / N), n = (int)(row - (long long)kb * N);
    const int c = cs ^ w_swz(n, cpr);
    const int k0 = kb * 64 + 8 * c;
    const float* w = t.w[l];
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = img == 0 ? w[(size_t)n * cin + k0 + j] : w[(size_t)(k0 + j) * cin + n];  // W[co][ci]
    t.out[l][local] = pack8(f);
  }
}

// ---------------------------------------------------------------------------------------------
// shared pieces
// ---------------------------------------------------------------------------------------------
// fragment of 8 channels (ch .. ch + 7 of the contraction side) from the raw 16-byte loads
template <int MODE>
__device__ __forceinline__ bf16x8 make_frag(uint4 r0, uint4 r1, const float* cA, int K, int ch) {
  float a[8], c0[8], c1[8];
  unpack8(r0, a);
  *reinterpret_cast<float4*>(c0) = *reinterpret_cast<const float4*>(cA + ch);
  *reinterpret_cast<float4*>(c0 + 4) = *reinterpret_cast<const float4*>(cA + ch + 4);
  *reinterpret_cast<float4*>(c1) = *reinterpret_cast<const float4*>(cA + K + ch);
  *reinterpret_cast<float4*>(c1 + 4) = *reinterpret_cast<const float4*>(cA + K + ch + 4);
  uint4 p;
  if constexpr (MODE == kFwd) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaf(c0[j], a[j], c1[j]);
    p = pack8(v);
    p.x = relu_pk(p.x); p.y = relu_pk(p.y); p.z = relu_pk(p.z); p.w = relu_pk(p.w);
  } else {
    float y[8], c2[8], v[8];
    unpack8(r1, y);
    *reinterpret_cast<float4*>(c2) = *reinterpret_cast<const float4*>(cA + 2 * K + ch);
    *reinterpret_cast<float4*>(c2 + 4) = *reinterpret_cast<const float4*>(cA + 2 * K + ch + 4);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = fmaf(c0[j], a[j], fmaf(c1[j], y[j], c2[j]));
    p = pack8(v);
  }
  return __builtin_bit_cast(bf16x8, p);
}

// constants of the contraction side into LDS: forward scale | shift; data gradient ga | gb | c0
template <int MODE>
__device__ __forceinline__ void fill_cA(float* cA, const float* bnA, int K, int tid, int nthreads) {
  for (int c = tid; c < K; c += nthreads) {
    if constexpr (MODE == kFwd) {
      const float sc = bnA[TTK_BN_SCALE * K + c];
      cA[c] = sc;
      cA[K + c] = fmaf(-sc, bnA[TTK_BN_MEAN * K + c], bnA[TTK_BN_BETA * K + c]);
    } else {
      const float ga = bnA[TTK_BN_GA * K + c], gb = bnA[TTK_BN_GB * K + c];
      cA[c] = ga;
      cA[K + c] = gb;
      cA[2 * K + c] = -ga * bnA[TTK_BN_GMEAN * K + c] - gb * bnA[TTK_BN_MEAN * K + c];
    }
  }
}
// ---------------------------------------------------------------------------------------------
// resident-weight kernel: N = 32 NB <= 128 output channels, K = 32 KH <= 256 contraction channels
// ---------------------------------------------------------------------------------------------
template <int MODE, int NB, int KH>
__global__ void __launch_bounds__(512) bc_gemm_e_k(const bf16_t* __restrict__ A0, const bf16_t* __restrict__ A1, const float* __restrict__ bnA,
                                                    const uint4* __restrict__ Wimg, bf16_t* __restrict__ out, const bf16_t* __restrict__ maskY,
                                                    const float* __restrict__ bnE, const float* __restrict__ pivot, float* __restrict__ part, int64_t M) {
  constexpr int K = 32 * KH, N = 32 * NB, CPR = KH == 1 ? 4 : 8, SS = KH == 1 ? 2 : 4, NKB = KH == 1 ? 1 : KH / 2, CBK = KH == 1 ? 32 : 64;
  constexpr int OC = NB == 1 ? 32 : 64, NCB = NB == 1 ? 1 : NB / 2, LPP = OC / 8, PPI = 64 / LPP, NI = 32 / PPI, BPC = OC / 32;
  extern __shared__ uint4 lds[];
  uint4* Wl = lds;                                             // N * K / 8 chunks
  float* cA = reinterpret_cast<float*>(lds + N * K / 8);       // [3][K]
  float* cE = cA + 3 * K;                                      // [3][N]
  uint4* stg = reinterpret_cast<uint4*>(cE + 3 * N);           // 8 waves x 4 OC chunks
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  for (int i = tid; i < N * K / 8; i += 512) Wl[i] = Wimg[i];
  fill_cA<MODE>(cA, bnA, K, tid, 512);
  fill_cE<MODE>(cE, pivot, bnE, N, 0, N, tid, 512);
  __syncthreads();

  uint4* mystg = stg + wave * (4 * OC);
  const int64_t groups = (M + 31) / 32, stride = (int64_t)gridDim.x * 8;
  int64_t g = (int64_t)blockIdx.x * 8 + wave;
  float s1[NCB][8], s2[NCB][8];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[cb][j] = s2[cb][j] = 0.f;
  uint4 rc0[SS], rc1[SS], rn0[SS], rn1[SS];
  auto load = [&](int64_t grp, int kb, uint4(&d0)[SS], uint4(&d1)[SS]) {
    int64_t pix = grp * 32 + r;
    pix = pix < M ? pix : M - 1;
    const size_t o = ((size_t)kb * M + pix) * CBK + 8 * h;
#pragma unroll
    for (int s = 0; s < SS; ++s) {
      d0[s] = ld16nt(A0 + o + 16 * s);
      if constexpr (MODE == kDgrad) d1[s] = ld16nt(A1 + o + 16 * s);
    }
  };
  // this lane's W fragment: row 32 nb + r, chunk 2 s + h (swizzled) of k64 block kb
  const int wrow = r * CPR, wsw = w_swz(r, CPR);
  if (g < groups) load(g, 0, rc0, rc1);
  for (; g < groups; g += stride) {
    const int64_t pix0 = g * 32;
    // the mask operand of the data gradient (same positions as the outputs this lane will store): block 0 is requested here, block cb + 1
    // when block cb is stored
    uint4 mk[2][NI];
    auto load_mask = [&](int cb, uint4(&d)[NI]) {
      if constexpr (MODE == kDgrad) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
          int64_t pix = pix0 + PPI * i + lane / LPP;
          pix = pix < M ? pix : M - 1;
          d[i] = ld16nt(maskY + ((size_t)cb * M + pix) * OC + 8 * (lane % LPP));
        }
      }
    };
    load_mask(0, mk[0]);
    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
#pragma unroll(NKB <= 2 ? NKB : 1)
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb + 1 < NKB) load(g, kb + 1, rn0, rn1);
      else if constexpr (MODE == kFwd && NKB == 1) load(g + stride < groups ? g + stride : g, 0, rn0, rn1);  // unconditional (the last group requests itself again): 32 -> 64 forward 67.5 -> 63.5 us
      else if (g + stride < groups) load(g + stride, 0, rn0, rn1);  // (the other shapes LOSE 8 - 17 % with the unconditional form: profiles/r05_bc_gemm_variants.txt)
#pragma unroll
      for (int s = 0; s < SS; ++s) {
        const bf16x8 fr = make_frag<MODE>(rc0[s], rc1[s], cA, K, kb * 64 + 16 * s + 8 * h);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const uint4 wv = Wl[(kb * N + 32 * nb) * CPR + wrow + ((2 * s + h) ^ wsw)];
          acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wv), fr, acc[nb], 0, 0, 0);
        }
      }
#pragma unroll
      for (int s = 0; s < SS; ++s) { rc0[s] = rn0[s]; rc1[s] = rn1[s]; }
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      if (cb + 1 < NCB) load_mask(cb + 1, mk[(cb + 1) & 1]);
      store_block<MODE, OC>(acc + cb * BPC, mystg, out + ((size_t)cb * M + pix0) * OC, mk[cb & 1], cE + cb * OC, N, pix0, M, s1[cb], s2[cb]);
    }
  }
  // ---- partial sums of the workgroup: lanes of one chunk, then the eight waves (fixed order)
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int off = LPP; off < 64; off <<= 1) {
        s1[cb][j] += __shfl_xor(s1[cb][j], off);
        s2[cb][j] += __shfl_xor(s2[cb][j], off);
      }
  __syncthreads();
  float* red = reinterpret_cast<float*>(stg);  // [8 waves][2][N]
  if (lane < LPP) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[(wave * 2 + 0) * N + cb * OC + 8 * lane + j] = s1[cb][j];
        red[(wave * 2 + 1) * N + cb * OC + 8 * lane + j] = s2[cb][j];
      }
  }
  __syncthreads();
  if (part)
    for (int i = tid; i < 2 * N; i += 512) {
      const int which = i / N, c = i - which * N;
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += red[(w * 2 + which) * N + c];
      part[(size_t)blockIdx.x * 2 * N + i] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// streamed-operand kernel: tiles of RT <= 256 pixels x 256 output channels; K a multiple of 128, N of 256.
//
// Both operands of a k64 step go through LDS: the weight slab of the column tile (256 rows x 128 B, copied as it lies in the image) and
// the TRANSFORMED activation tile (256 pixels x 64 channels bf16).  Every thread stages four FIXED 16-byte chunk columns of the
// activations (chunk = its 8 channels: the BatchNorm constants of a step are two or three LDS reads per thread, not per fragment) from
// whole 128-byte lines, one step ahead in registers.  The eight waves form 2 (channel halves) x 4 (pixel quarters): a wave multiplies
// 128 channels x 64 pixels = 4 x 2 blocks, so a k16 sub-step is 4 weight + 2 activation fragment reads for 8 MFMAs (the first form -
// fragments loaded straight from global memory, 256 channels x 32 pixels per wave - read 8 + 4 constants' worth per 8 MFMAs and held the
// LDS pipe at 75 %: 55 us for the 512 x 512 forward).  Two steps per loop iteration with the register sets swapping roles, every load
// unconditional (clamped indices): hipcc's counted vmcnt waits stay exact.
// ---------------------------------------------------------------------------------------------
// NCT: output channels per tile (256; 128 for the one layer with N = 128 and K = 256 - its data gradient ran at 0.29 of 8 TB/s in the resident-weight form)
template <int MODE, int NCT>
__global__ void __launch_bounds__(512) bc_gemm_l_k(const bf16_t* __restrict__ A0, const bf16_t* __restrict__ A1, const float* __restrict__ bnA,
                                                    const uint4* __restrict__ Wimg, bf16_t* __restrict__ out, const bf16_t* __restrict__ maskY,
                                                    const float* __restrict__ bnE, const float* __restrict__ pivot, float* __restrict__ part, int64_t M,
                                                    int K, int N, int RT, int nrt, int ncol) {
  extern __shared__ uint4 lds[];
  constexpr int WCH = NCT * 8, WPT = WCH / 512, NBW = NCT / 64, NCB = NCT / 128;  // weight chunks per stage / per thread; 32-channel blocks / 64-channel blocks per wave
  uint4* wring = lds;                                           // [2][NCT rows x 8 chunks]
  uint4* aring = lds + 2 * WCH;                                 // [2][256 pixels x 8 chunks]; after the loop: 8 waves x 256 chunks of store tiles
  float* cE = reinterpret_cast<float*>(lds + 2 * WCH + 2 * 2048);  // [3][NCT]
  float* cA = cE + 3 * NCT;                                     // [3][K]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int wc = wave & 1, wp = wave >> 1;
  // blocks b and b + 8 share an XCD (round-robin dispatch): the column tiles of one row tile sit there together, the second one finds
  // the activation rows in that XCD's L2
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int ct = jj % ncol, rt = (jj / ncol) * 8 + xcd;
  if (rt >= nrt) return;
  const int n0 = ct * NCT;
  const int64_t p0 = (int64_t)rt * RT, pend = (p0 + RT < M) ? p0 + RT : M;
  const int nkb = K / 64, klast = nkb - 1;
  // staging role: chunk column `oct` (channels 8 oct .. + 7 of the k64 block) of pixels (tid >> 3) + 64 i
  const int oct = tid & 7;
  size_t soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int64_t px = p0 + (tid >> 3) + 64 * i;
    px = px < pend ? px : pend - 1;
    soff[i] = (size_t)px * 64 + 8 * oct;
  }
  // Register chunks of the NEXT step's operands: chunk i (activation pixels (tid >> 3) + 64 i, weight chunk tid + 512 i) is transformed and
  // stored behind the MFMAs of sub-step i of the current step, and reloaded at once for the step after next - a ring of four slots per
  // operand, every load unconditional (clamped to the last slab: the surplus copies are never read).
  u32x4 rx0[4], rx1[4], wr[4];  // (wr: WPT of them are used)  // (ext-vector registers: arrays of the HIP struct type captured by the lambdas went to scratch)
  const u32x4* Wv = reinterpret_cast<const u32x4*>(Wimg);
  auto load_chunk = [&](int kb, int i) {
    const size_t o = (size_t)kb * M * 64 + soff[i];  // (plain loads: the other column tiles of these pixels read the same lines from L2)
    rx0[i] = *reinterpret_cast<const u32x4*>(A0 + o);
    if constexpr (MODE == kDgrad) rx1[i] = *reinterpret_cast<const u32x4*>(A1 + o);
    if (i < WPT) wr[i] = Wv[((size_t)kb * N + n0) * 8 + tid + 512 * i];
  };
  auto store_chunk = [&](int slot, int kb, int i) {
    const int px = (tid >> 3) + 64 * i;
    const bf16x8 fr = make_frag<MODE>(make_uint4(rx0[i].x, rx0[i].y, rx0[i].z, rx0[i].w), make_uint4(rx1[i].x, rx1[i].y, rx1[i].z, rx1[i].w), cA, K, kb * 64 + 8 * oct);
    aring[slot * 2048 + px * 8 + (oct ^ (px & 7))] = __builtin_bit_cast(uint4, fr);
    if (i < WPT) reinterpret_cast<u32x4*>(wring)[slot * WCH + tid + 512 * i] = wr[i];
  };
  const int wrow = ((NCT / 2) * wc + r) * 8, wsw = (r >> 1) & 7;
  const int arow = (64 * wp + r) * 8, asw = r & 7;
  f32x16 acc[2][NBW];  // [pixel group][channel block]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][nb][e] = 0.f;
  // (the first step's operands are requested before the constants; measured: no difference to the other order, profiles/r05_bc_gemm_variants.txt)
#pragma unroll
  for (int i = 0; i < 4; ++i) load_chunk(0, i);
  fill_cA<MODE>(cA, bnA, K, tid, 512);
  fill_cE<MODE>(cE, pivot, bnE, N, n0, NCT, tid, 512);
  __syncthreads();  // constants are in LDS
#pragma unroll
  for (int i = 0; i < 4; ++i) store_chunk(0, 0, i);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 4; ++i) load_chunk(1, i);  // (the loop is entered with the loads pending in the order it leaves them)
  __syncthreads();
  for (int kb = 0; kb < nkb; ++kb) {
    const int slot = kb & 1;
    const uint4* Ws = wring + slot * WCH;
    const uint4* As = aring + slot * 2048;
    const int knext = min(kb + 1, klast), knn = min(kb + 2, klast);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      uint4 av[2], wv[NBW];
#pragma unroll
      for (int j = 0; j < 2; ++j) av[j] = (TTK_BC_GDBG & 8) ? make_uint4(kb, s4, j, lane) : As[arow + 256 * j + ((2 * s4 + h) ^ asw)];
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb) wv[nb] = (TTK_BC_GDBG & 8) ? make_uint4(kb, s4, nb, lane) : Ws[wrow + 256 * nb + ((2 * s4 + h) ^ wsw)];
#pragma unroll
      for (int nb = 0; nb < NBW; ++nb)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (TTK_BC_GDBG & 1) { acc[j][nb][0] += __uint_as_float(wv[nb].x ^ av[j].y ^ wv[nb].z ^ av[j].w); continue; }
          acc[j][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wv[nb]), __builtin_bit_cast(bf16x8, av[j]), acc[j][nb], 0, 0, 0);
        }
      // behind them (the matrix pipe leaves three quarters of the issue slots to vector instructions): chunk s4 of the next step into the
      // other slot (last read a step ago: every wave has passed a barrier since), and its registers refilled for the step after
      if (!(TTK_BC_GDBG & 4)) store_chunk(slot ^ 1, knext, s4);
      else acc[0][0][1] += __uint_as_float(rx0[s4].x ^ wr[s4].y);
      if (!(TTK_BC_GDBG & 2)) load_chunk(knn, s4);
    }
    __syncthreads();
  }
  // ---- epilogue: the wave's 2 channel blocks of 64 x 2 pixel groups of 32 through its LDS tile (the activation ring is free: barrier above)
  uint4* mystg = aring + wave * 256;
  float* red = reinterpret_cast<float*>(wring);  // [8 waves][NCB blocks][2][64]
  const int o8 = lane & 7;
  uint4 mk[2][4];
  auto load_mask = [&](int q4, uint4(&d)[4]) {  // q4 = 2 * (local channel block) + pixel group
    if constexpr (MODE == kDgrad && !(TTK_BC_GDBG & 16)) {
      const int cb = n0 / 64 + NCB * wc + (q4 >> 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int64_t p = p0 + 64 * wp + 32 * (q4 & 1) + 8 * i + (lane >> 3);
        p = p < pend ? p : pend - 1;
        d[i] = ld16nt(maskY + ((size_t)cb * M + p) * 64 + 8 * o8);
      }
    }
  };
  load_mask(0, mk[0]);
#pragma unroll
  for (int cbl = 0; cbl < NCB; ++cbl) {
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    const int cb = n0 / 64 + NCB * wc + cbl;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int q4 = 2 * cbl + j;
      if (q4 + 1 < 2 * NCB) load_mask(q4 + 1, mk[(q4 + 1) & 1]);
      const int64_t g0 = p0 + 64 * wp + 32 * j;
      if ((TTK_BC_GDBG & 16) && acc[j][2 * cbl][0] != 12345.f) continue;  // (timing only: no epilogue)
      if (g0 < pend)
        store_block<MODE, 64>(acc[j] + 2 * cbl, mystg, out + ((size_t)cb * M + g0) * 64, mk[q4 & 1], cE + 64 * (NCB * wc + cbl), NCT, g0, pend, s1, s2);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int off = 8; off < 64; off <<= 1) {
        s1[j] += __shfl_xor(s1[j], off);
        s2[j] += __shfl_xor(s2[j], off);
      }
    if (lane < 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[((wave * NCB + cbl) * 2 + 0) * 64 + 8 * lane + j] = s1[j];
        red[((wave * NCB + cbl) * 2 + 1) * 64 + 8 * lane + j] = s2[j];
      }
    }
  }
  __syncthreads();
  if (part && tid < 2 * NCT) {  // tid = which * NCT + column of the tile; column c: channel half c / (NCT / 2), local 64-channel block, channel in it
    const int which = tid / NCT, c = tid % NCT, cwc = c / (NCT / 2), cbl = (c % (NCT / 2)) >> 6, cc = c & 63;
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) a += red[(((2 * q + cwc) * NCB + cbl) * 2 + which) * 64 + cc];
    part[((size_t)rt * 2 + which) * N + n0 + c] = a;
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static bool pw_shape_ok(int Cin, int Cout) {
  auto p2 = [](int v) { return v >= 32 && v <= 1024 && (v & (v - 1)) == 0; };
  return p2(Cin) && p2(Cout);
}
// E: N <= 128 and K <= 256; else L needs N % 256 == 0 and K % 64 == 0
static bool is_e(int K, int N) { return N <= 128 && K <= 256; }
static bool is_l(int K, int N) { return N % 256 == 0 && K % 128 == 0; }
static bool is_l128(int K, int N) { return N == 128 && K == 256; }  // (K = 128, N = 128 - the 128 -> 128 forward - is faster in the resident-weight form: 68 vs 90 us)  // (takes precedence over the resident-weight form for this shape, in both modes: the part rows depend on it)

struct LPlan { int RT, nrt, ncol, grid; };
static LPlan l_plan(int64_t M, int N, int nct = 256) {
  LPlan p;
  p.ncol = N / nct;
  const int64_t t256 = ceil_div(M, 256);
  const int64_t rounds = ceil_div(t256 * p.ncol, 256);      // rounds of the 256 CUs with full 256-pixel tiles
  int64_t target = rounds * 256 / p.ncol;                   // row tiles that fill those rounds
  if (target < 1) target = 1;
  int64_t RT = ceil_div(ceil_div(M, target), 8) * 8;
  if (RT > 256) RT = 256;
  if (RT < 8) RT = 8;
  p.RT = (int)RT;
  p.nrt = (int)ceil_div(M, RT);
  p.grid = (int)(ceil_div(p.nrt, 8) * 8 * p.ncol);
  return p;
}
static int e_grid(int64_t M) {
  const int64_t groups = ceil_div(M, 32);
  int64_t g = ceil_div(groups, 8);
  if (g > 256) g = 256;
  return (int)g;
}
static size_t e_lds_bytes(int K, int N) {
  const int OC = N < 64 ? 32 : 64;
  const size_t stg = (size_t)8 * 4 * OC * 16, red = (size_t)8 * 2 * N * 4;
  return (size_t)N * K * 2 + (size_t)3 * K * 4 + (size_t)3 * N * 4 + (stg > red ? stg : red);
}
static size_t l_lds_bytes(int K, int nct = 256) { return (size_t)(2 * nct * 8 + 2 * 2048) * 16 + (size_t)3 * nct * 4 + (size_t)3 * K * 4; }

template <int MODE>
static int launch_gemm(const bf16_t* A0, const bf16_t* A1, const float* bnA, const uint4* Wimg, bf16_t* out, const bf16_t* maskY, const float* bnE,
                       const float* pivot, float* part, int64_t M, int K, int N, hipStream_t st) {
  if (is_l128(K, N)) {
    const LPlan p = l_plan(M, N, 128);
    (void)allow_big_lds<bc_gemm_l_k<MODE, 128>>();
    hipLaunchKernelGGL((bc_gemm_l_k<MODE, 128>), dim3(p.grid), dim3(512), l_lds_bytes(K, 128), st, A0, A1, bnA, Wimg, out, maskY, bnE, pivot, part, M, K, N, p.RT, p.nrt, p.ncol);
    return 0;
  }
  if (is_e(K, N)) {
    const int grid = e_grid(M);
    const size_t sm = e_lds_bytes(K, N);
#define TTK_BC_E(NB_, KH_)                                                                                                                        \
  if (N == 32 * NB_ && K == 32 * KH_) {                                                                                                           \
    (void)allow_big_lds<bc_gemm_e_k<MODE, NB_, KH_>>();                                                                                                 \
    hipLaunchKernelGGL((bc_gemm_e_k<MODE, NB_, KH_>), dim3(grid), dim3(512), sm, st, A0, A1, bnA, Wimg, out, maskY, bnE, pivot, part, M);             \
    return 0;                                                                                                                                     \
  }
    TTK_BC_E(2, 1) TTK_BC_E(4, 2) TTK_BC_E(4, 4)            // forward: 32 -> 64, 64 -> 128, 128 -> 128
    TTK_BC_E(1, 2) TTK_BC_E(2, 4)                           // data gradient of 32 -> 64, 64 -> 128 (and 4, 4: 128 -> 128); K = 256 runs the streamed kernel
#undef TTK_BC_E
    return -2;
  }
  if (!is_l(K, N)) return -2;
  const LPlan p = l_plan(M, N);
  (void)allow_big_lds<bc_gemm_l_k<MODE, 256>>();
  hipLaunchKernelGGL((bc_gemm_l_k<MODE, 256>), dim3(p.grid), dim3(512), l_lds_bytes(K), st, A0, A1, bnA, Wimg, out, maskY, bnE, pivot, part, M, K, N, p.RT, p.nrt, p.ncol);
  return 0;
}

}  // namespace bc
}  // namespace ttk

using namespace ttk;
using namespace ttk::bc;

extern "C" {

size_t ttk_bc_prepared_bytes(int Cin, int Cout) { return pw_shape_ok(Cin, Cout) ? (size_t)2 * Cin * Cout * 2 : 0; }

int ttk_bc_prepare_weights(int n, const float* const* w, const int* cin, const int* cout, void* const* prepared, ttk_stream_t stream) {
  TTK_REQUIRE(n >= 1 && n <= kMaxPrep && w && cin && cout && prepared, "bc_prepare_weights: 1..%d layers", kMaxPrep);
  PrepTable t;
  t.n = n;
  long long total = 0;
  for (int i = 0; i < n; ++i) {
    TTK_REQUIRE(w[i] && prepared[i] && pw_shape_ok(cin[i], cout[i]), "bc_prepare_weights: layer %d: null pointer or unsupported shape %d -> %d", i, cin[i], cout[i]);
    t.w[i] = w[i];
    t.out[i] = reinterpret_cast<uint4*>(prepared[i]);
    t.cin[i] = cin[i];
    t.cout[i] = cout[i];
    t.start[i] = total;
    total += (long long)2 * cin[i] * cout[i] / 8;
  }
  t.start[n] = total;
  int grid = (int)ceil_div(total, 256);
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(bc_prepare_k, dim3(grid), dim3(256), 0, (hipStream_t)stream, t);
  TTK_LAUNCH_CHECK("bc_prepare_weights");
}

int ttk_bc_partial_rows_pw(int64_t M, int K, int Nout) {
  if (M < 1 || !pw_shape_ok(K, Nout)) return -1;
  if (is_l128(K, Nout)) return l_plan(M, Nout, 128).nrt;
  if (is_e(K, Nout)) return e_grid(M);
  if (is_l(K, Nout)) return l_plan(M, Nout).nrt;
  return -1;
}

int ttk_bc_pw_fwd(const void* ydw, const float* bn_dw, const void* wprep, void* y, float* part, const float* pivot, int64_t M, int Cin, int Cout,
                  ttk_stream_t stream) {
  TTK_REQUIRE(ydw && bn_dw && wprep && y, "bc_pw_fwd: null pointer");
  TTK_REQUIRE(M >= 1 && M < ((int64_t)1 << 31) && pw_shape_ok(Cin, Cout) && (is_e(Cin, Cout) || is_l(Cin, Cout)), "bc_pw_fwd: unsupported shape M=%lld %d -> %d",
              (long long)M, Cin, Cout);
  const int rc = launch_gemm<kFwd>((const bf16_t*)ydw, nullptr, bn_dw, (const uint4*)wprep, (bf16_t*)y, nullptr, nullptr, pivot, part, M, Cin, Cout, (hipStream_t)stream);
  TTK_REQUIRE(rc == 0, "bc_pw_fwd: no kernel for %d -> %d", Cin, Cout);
  TTK_LAUNCH_CHECK("bc_pw_fwd");
}

int ttk_bc_pw_bwd_data(const void* g, const void* y, const float* bn_pw, const void* wprep, const void* ydw, const float* bn_dw, void* g_dw,
                       float* part, int64_t M, int Cin, int Cout, ttk_stream_t stream) {
  TTK_REQUIRE(g && y && bn_pw && wprep && ydw && bn_dw && g_dw, "bc_pw_bwd_data: null pointer");
  TTK_REQUIRE(M >= 1 && M < ((int64_t)1 << 31) && pw_shape_ok(Cin, Cout) && (is_e(Cout, Cin) || is_l(Cout, Cin)), "bc_pw_bwd_data: unsupported shape M=%lld %d -> %d",
              (long long)M, Cin, Cout);
  const uint4* img = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(wprep) + (size_t)Cin * Cout * 2);  // image 1: rows = Cin, k = Cout
  const int rc = launch_gemm<kDgrad>((const bf16_t*)g, (const bf16_t*)y, bn_pw, img, (bf16_t*)g_dw, (const bf16_t*)ydw, bn_dw, nullptr, part, M, Cout, Cin,
                                     (hipStream_t)stream);
  TTK_REQUIRE(rc == 0, "bc_pw_bwd_data: no kernel for %d -> %d", Cin, Cout);
  TTK_LAUNCH_CHECK("bc_pw_bwd_data");
}

}  // extern "C"
