// Backward of the first pointwise convolutions (32 -> 64, 64 -> 128 and - second half of this file - 128 -> 128 channels; DepthWiseBlock.conv_sep + bn_sep,
// backbones/mobilenet_v1.py:67-68,82-84) as ONE kernel: data gradient and weight gradient from one read of the operands.
//
// These layers have the largest activations of the network and are HBM-bound.  As two kernels (pwconv.hip: pw_gemm_k
// in data-gradient mode, pw_wgrad_k) the gradient g and the raw output y of the convolution (2 x M x Cout floats) and the
// raw depthwise output ydw (M x Cin) are read twice: 1.38 + 1.66 GB for the 32 -> 64 layer at B = 512, against 1.66 GB
// for this kernel (g, y, ydw read once, g_dw written).  Both products need the same two tiles,
//     dy = ga*(g - gmean) + gb*(y - mean_pw)   [rows][Cout]      and      yc = ydw - mean_dw   [rows][Cin],
//     g_dw = (dy W) * [scale*yc + beta > 0]                 (data gradient through the ReLU of bn_dw)
//     dW  += dy^T relu(scale*yc + beta)                      (weight gradient)
// which a workgroup stages once per 64 rows in LDS.  v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation - the
// arithmetic of the two kernels it replaces.
//
// One workgroup = 8 waves, persistent over 64-row tiles.  Waves 4-7 produce (16-byte loads of g, y, ydw one tile ahead, the
// BatchNorm forms, float4 LDS writes into a double-buffered stage); waves 0-3 multiply: every wave owns whole 32x32 output
// tiles - data-gradient tiles [row tile][ci tile] (contraction over Cout) whose masked result goes straight from the
// accumulators to global memory together with the BatchNorm-backward sums of bn_dw, and weight-gradient tiles
// [co tile][ci tile] (contraction over the 64 rows) that stay in registers until the workgroup has seen all its rows.
// The MFMA's two k slots of a lane half h take k = 8q + 4h + i, i = 0..3, for four consecutive instructions, so that the
// data gradient's fragments are ds_read_b128 (4 k-values per read) - any k order is a valid contraction order.
#include "ttk_common.h"
#include "conv_geom.h"

namespace ttk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kFusedBM = 64;

template <int CIN, int COUT>
struct FusedShape {
  static constexpr int LDY = COUT + 4, LDC = CIN + 4, LDW = COUT + 4;
  static constexpr int NTC = CIN / 32, MTO = COUT / 32;
  static constexpr int DG = 2 * NTC;    // data-gradient jobs: 2 row tiles x ci tiles
  static constexpr int WG = MTO * NTC;  // weight-gradient jobs
  static constexpr size_t smem_floats = (size_t)CIN * LDW + 2 * kFusedBM * LDY + 2 * kFusedBM * LDC + 4 * 2 * CIN;
};

template <int CIN, int COUT>
__global__ void __launch_bounds__(512) pw_bwd_fused_k(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ bn_pw,
                                                       const float* __restrict__ w, const float* __restrict__ ydw,
                                                       const float* __restrict__ bn_dw, float* __restrict__ g_dw, float* __restrict__ dW,
                                                       float* __restrict__ wpartial, float* __restrict__ part, int64_t M, int ntiles) {
  using S = FusedShape<CIN, COUT>;
  constexpr int BM = kFusedBM, LDY = S::LDY, LDC = S::LDC, LDW = S::LDW, NTC = S::NTC, DG = S::DG, WG = S::WG;
  static_assert((CIN == 32 && COUT == 64) || (CIN == 64 && COUT == 128), "shapes of the first two pointwise layers");
  // jobs per consumer wave: 32 -> 64: waves 0,1 one data-gradient tile each, waves 2,3 one weight-gradient tile each;
  // 64 -> 128: every wave one data-gradient tile and two weight-gradient tiles
  constexpr int DGW = DG >= 4 ? DG / 4 : 1, WGW = WG >= 4 ? WG / 4 : 1;
  static_assert(DGW == 1, "one data-gradient tile per consumer wave");
  __shared__ __attribute__((aligned(16))) float smem[S::smem_floats];
  float* Wt = smem;                    // [CIN][LDW]: Wt[ci][co] = w[co][ci]
  float* DyS = Wt + CIN * LDW;         // [2][BM][LDY]
  float* YcS = DyS + 2 * BM * LDY;     // [2][BM][LDC]
  float* red = YcS + 2 * BM * LDC;     // [4][2][CIN]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int r32 = lane & 31, h = lane >> 5;
  // Weight-gradient jobs: on the consumer waves (32 -> 64: waves 2 and 3 one tile each, the kernel is at its HBM floor;
  // 64 -> 128: two tiles per wave beside its data-gradient tile - 128 fp32 MFMAs = 8192 cycles per SIMD and 64 rows, which
  // together with the scalar LDS fragment reads makes that layer matrix-bound: 263 us against 165 us of HBM time).  Moving
  // them to the producer waves (WG_PROD) does not help - 286 us: the matrix pipe belongs to the SIMD, not to the wave.
  constexpr bool WG_PROD = false;
  const int wgw = WG_PROD ? wave - 4 : wave;  // this wave's index among the waves that carry weight-gradient jobs
  const bool has_wg = WG_PROD ? producer : (!producer && (WG >= 4 || wave >= 4 - WG));
  const int wj0 = WG >= 4 ? wgw * WGW : wgw - (4 - WG);
  f32x16 wacc[WGW];
  float wsc[WGW], wbe[WGW];
#pragma unroll
  for (int j = 0; j < WGW; ++j) {
#pragma unroll
    for (int e = 0; e < 16; ++e) wacc[j][e] = 0.f;
    const int col = ((has_wg ? wj0 + j : 0) % NTC) * 32 + r32;
    wsc[j] = bn_dw[TTK_BN_SCALE * CIN + col];
    wbe[j] = bn_dw[TTK_BN_BETA * CIN + col];
  }
  // between(k) runs after the k-th group of four MFMAs (k < WGW * BM / 8): the data gradient's epilogue rides there, in the
  // issue slots the matrix instructions leave free
  auto wgrad_jobs = [&](int st, auto&& between) {
    const float* Dy = DyS + st * BM * LDY;
    const float* Yc = YcS + st * BM * LDC;
#pragma unroll
    for (int j = 0; j < WGW; ++j) {
      const int mt = (wj0 + j) / NTC, nt = (wj0 + j) % NTC;
      const float* ap = Dy + (4 * h) * LDY + mt * 32 + r32;
      const float* bp = Yc + (4 * h) * LDC + nt * 32 + r32;
      // fragments of the next four MFMAs are requested before the current four issue (the compiler's own order waited for
      // every ds_read right in front of its MFMA: the matrix pipe idled for an LDS latency per instruction pair)
      float a[2][4], b[2][4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[0][i] = ap[i * LDY]; b[0][i] = bp[i * LDC]; }
#pragma unroll
      for (int q = 0; q < BM / 8; ++q) {
        const int cur = q & 1;
        if (q + 1 < BM / 8) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { a[cur ^ 1][i] = ap[(8 * (q + 1) + i) * LDY]; b[cur ^ 1][i] = bp[(8 * (q + 1) + i) * LDC]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
          wacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], fmaxf(fmaf(wsc[j], b[cur][i], wbe[j]), 0.f), wacc[j], 0, 0, 0);
        between(j * (BM / 8) + q);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  for (int i = tid; i < CIN * COUT; i += 512) {
    const int co = i / CIN, ci = i - co * CIN;
    Wt[ci * LDW + co] = w[i];
  }

  if (producer) {
    const int pt = tid - 256;
    constexpr int QY = COUT / 4, QC = CIN / 4;        // float4 per row
    constexpr int NY = BM * QY / 256, NC = BM * QC / 256;  // float4 per thread and tile
    const int cy = pt % QY, cc = pt % QC;             // (256 is a multiple of QY and QC: the column quad of a thread is fixed)
    const float4 ga = ld4(bn_pw + TTK_BN_GA * COUT + 4 * cy), gb = ld4(bn_pw + TTK_BN_GB * COUT + 4 * cy);
    const float4 gmean = ld4(bn_pw + TTK_BN_GMEAN * COUT + 4 * cy), ymean = ld4(bn_pw + TTK_BN_MEAN * COUT + 4 * cy);
    const float4 dmean = ld4(bn_dw + TTK_BN_MEAN * CIN + 4 * cc);
    float4 rg[NY], ry[NY], rc[NC];
    auto load = [&](int t) {
      const int64_t m0 = (int64_t)t * BM;
#pragma unroll
      for (int i = 0; i < NY; ++i) {
        const int64_t row = m0 + (pt + 256 * i) / QY;
        const int64_t rcl = row < M ? row : M - 1;
        rg[i] = ld4nt(g + act_off(rcl, 4 * cy, M));  // activations: channel blocks (ttk_common.h)
        ry[i] = ld4nt(y + act_off(rcl, 4 * cy, M));
      }
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int64_t row = m0 + (pt + 256 * i) / QC;
        const int64_t rcl = row < M ? row : M - 1;
        rc[i] = ld4(ydw + act_off(rcl, 4 * cc, M));  // (read again by the depthwise backward: cached)
      }
    };
    auto store = [&](int t, int st) {
      const int64_t m0 = (int64_t)t * BM;
#pragma unroll
      for (int i = 0; i < NY; ++i) {
        const int r = (pt + 256 * i) / QY;
        float4 v = fma4(ga, sub4(rg[i], gmean), mul4(gb, sub4(ry[i], ymean)));
        if (m0 + r >= M) v = f4(0.f);  // rows past the end contribute nothing to the weight gradient
        st4(DyS + (st * BM + r) * LDY + 4 * cy, v);
      }
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int r = (pt + 256 * i) / QC;
        st4(YcS + (st * BM + r) * LDC + 4 * cc, sub4(rc[i], dmean));
      }
    };
    int t = blockIdx.x;
    if (t < ntiles) {
      load(t);
      store(t, 0);
      if (t + (int)gridDim.x < ntiles) load(t + gridDim.x);
    }
    __syncthreads();
    for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
      const int tn = t + gridDim.x;
      if (tn < ntiles) {
        store(tn, (it + 1) & 1);
        if (tn + (int)gridDim.x < ntiles) load(tn + gridDim.x);  // lands under the MFMAs below
      }
      if constexpr (WG_PROD) wgrad_jobs(it & 1, [](int) {});
      __syncthreads();
    }
  } else {
    // this wave's data-gradient job
    const bool has_dg = DG >= 4 || wave < DG;
    const int dj0 = DG >= 4 ? wave * DGW : wave;
    float s1[DGW], s2[DGW], dsc[DGW], dbe[DGW];
#pragma unroll
    for (int j = 0; j < DGW; ++j) {
      const int col = ((dj0 + j) % NTC) * 32 + r32;
      s1[j] = 0.f; s2[j] = 0.f;
      dsc[j] = has_dg ? bn_dw[TTK_BN_SCALE * CIN + col] : 0.f;
      dbe[j] = has_dg ? bn_dw[TTK_BN_BETA * CIN + col] : 0.f;
    }
    __syncthreads();  // Wt and stage 0
    int it = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, ++it) {
      const int st = it & 1;
      const float* Dy = DyS + st * BM * LDY;
      const float* Yc = YcS + st * BM * LDC;
      const int64_t m0 = (int64_t)t * BM;
      if (has_dg) {
#pragma unroll
        for (int j = 0; j < DGW; ++j) {
          const int mt = (dj0 + j) / NTC, nt = (dj0 + j) % NTC;
          f32x16 acc;
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[e] = 0.f;
          const float* ap = Dy + (mt * 32 + r32) * LDY + 4 * h;
          const float* bp = Wt + (nt * 32 + r32) * LDW + 4 * h;
          float4 a4[2], b4[2];
          a4[0] = ld4(ap); b4[0] = ld4(bp);
#pragma unroll
          for (int q = 0; q < COUT / 8; ++q) {
            const int cur = q & 1;
            if (q + 1 < COUT / 8) { a4[cur ^ 1] = ld4(ap + 8 * (q + 1)); b4[cur ^ 1] = ld4(bp + 8 * (q + 1)); }  // one group ahead
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cur].x, b4[cur].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cur].y, b4[cur].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cur].z, b4[cur].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cur].w, b4[cur].w, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
          // accumulator element e of lane (r32, h): row (e & 3) + 8 (e >> 2) + 4 h of the tile, column r32
          const int col = nt * 32 + r32;
          auto epi = [&](int e) {
            const int row = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
            const float yc = Yc[row * LDC + col];
            const float out = fmaf(dsc[j], yc, dbe[j]) > 0.f ? acc[e] : 0.f;  // ReLU mask of bn_dw
            if (m0 + row < M) {
              g_dw[act_off(m0 + row, col, M)] = out;  // 32 lanes = 128 contiguous bytes (one row of channel block nt)
              s1[j] += out;
              s2[j] = fmaf(out, yc, s2[j]);
            }
          };
          if constexpr (!WG_PROD && WG >= 4 && WGW * (BM / 8) == 16) {
            wgrad_jobs(st, epi);  // 16 groups of weight-gradient MFMAs, one epilogue element behind each
          } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) epi(e);
          }
        }
      }
      if constexpr (!WG_PROD && !(WG >= 4 && WGW * (BM / 8) == 16))
        if (has_wg) wgrad_jobs(st, [](int) {});
      __syncthreads();
    }
    // ---- BatchNorm-backward sums of bn_dw: fold the lane halves, then the waves of a ci tile in a fixed order
#pragma unroll
    for (int j = 0; j < DGW; ++j) {
      s1[j] += __shfl_xor(s1[j], 32);
      s2[j] += __shfl_xor(s2[j], 32);
    }
    if (has_dg && h == 0) {
      static_assert(DGW == 1, "one data-gradient tile per wave");
      red[(wave * 2 + 0) * CIN + (dj0 % NTC) * 32 + r32] = s1[0];
      red[(wave * 2 + 1) * CIN + (dj0 % NTC) * 32 + r32] = s2[0];
    }
  }
  // ---- the workgroup's share of dW: one atomicAdd per element (or, deterministic mode, its row of wpartial)
  if (has_wg) {
#pragma unroll
    for (int j = 0; j < WGW; ++j) {
      const int mt = (wj0 + j) / NTC, nt = (wj0 + j) % NTC;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = mt * 32 + (e & 3) + 8 * (e >> 2) + 4 * h, ci = nt * 32 + r32;
        if (wpartial) wpartial[(size_t)blockIdx.x * COUT * CIN + co * CIN + ci] = wacc[j][e];
        else atomicAdd(dW + co * CIN + ci, wacc[j][e]);
      }
    }
  }
  __syncthreads();
  if (part && tid < 2 * CIN) {
    const int which = tid / CIN, c = tid % CIN, nt = c / 32;
    float a = 0.f;
    for (int wv = 0; wv < (DG >= 4 ? 4 : DG); ++wv)
      if (wv % NTC == nt) a += red[(wv * 2 + which) * CIN + c];  // the waves whose tile covers channel c, in wave order
    part[(size_t)blockIdx.x * 2 * CIN + which * CIN + c] = a;
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// The 128 -> 128 layer: the same fusion on the fp16 matrix pipe (fp32 MFMA would bind: 8.4 MFLOP per 64 rows).  Arithmetic
// of pwconv_f16.hip: operands scaled by powers of two taken from their bounds (dy: TTK_AUX_DY_BOUND of bn_pw, a:
// TTK_AUX_ACT_BOUND of bn_dw, W: the |w| maximum in the prepared block), cut into two fp16 pieces, three products.
//
// What makes it fit: the WEIGHTS live in registers.  Consumer wave w owns input-channel tile w (32 of 128) for both products;
// its data-gradient B fragments (W^T planes of 32 ci x 128 co: 8 k16 steps x 2 planes) are 64 VGPRs loaded once from the
// prepared data-gradient operand, so LDS only holds the per-stage tiles of 32 rows, double-buffered (2 x 65 KB):
//   DyR  dy row-major   [plane][32 rows][128 co]  - data gradient, A operand (k = co); 16-byte chunk XOR (row & 15)
//   DyT  dy transposed  [k16][plane][128 co][16 rows] - weight gradient, A operand (k = rows); layout of pwconv_f16.hip
//   AT   a  transposed  [k16][plane][128 ci][16 rows] - weight gradient, B operand
//   Yc   ydw - mean, fp32 [32][132] - ReLU mask and the BatchNorm-backward sums of the epilogue
// The producers load 4 rows x 4 channels per thread (as the weight-gradient kernels do) and write both layouts of dy from the
// same registers.  Per 32 rows a consumer wave issues 24 (data gradient) + 24 (weight gradient) MFMAs = 1536 cycles against
// 64 KB of operands through the CU (6400 cycles): memory-bound, as it should be.
typedef _Float16 fz16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 fz16x2 __attribute__((ext_vector_type(2)));
typedef float fz32x2 __attribute__((ext_vector_type(2)));
typedef float fz32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int fz_swz(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 3) & 1)) << 4); }
// h / l fp16 pieces of 4 (already scaled) values: 8 bytes each
__device__ __forceinline__ void fz_split(fz32x4 v, uint2& h, uint2& l) {
  const fz16x2 h01 = __builtin_convertvector(fz32x2{v.x, v.y}, fz16x2), h23 = __builtin_convertvector(fz32x2{v.z, v.w}, fz16x2);
  const fz32x2 f01 = __builtin_convertvector(h01, fz32x2), f23 = __builtin_convertvector(h23, fz32x2);
  const fz16x2 l01 = __builtin_convertvector(fz32x2{v.x - f01.x, v.y - f01.y}, fz16x2);
  const fz16x2 l23 = __builtin_convertvector(fz32x2{v.z - f23.x, v.w - f23.y}, fz16x2);
  h = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
  l = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}

constexpr int kF16BM = 32;
template <int CIN, int COUT>
struct F16Shape {
  static constexpr int RowPlane = kF16BM * COUT * 2;  // bytes of one plane of DyR
  static constexpr int TPlaneY = COUT * 32;           // bytes of one plane of one k16 stage of DyT
  static constexpr int TPlaneA = CIN * 32;            // ... of AT
  static constexpr int Ldc = CIN + 4;
  static constexpr int Stage = 2 * RowPlane + 2 * 2 * TPlaneY + 2 * 2 * TPlaneA + kF16BM * Ldc * 4;  // DyR + DyT + AT + Yc
  static constexpr int NCI = CIN / 32, NCO = COUT / 32;  // 32-channel tiles
  static constexpr int GROUPS = 4 / NCI;                 // consumer-wave groups that share the ci tiles
  static constexpr int COW = NCO / GROUPS;               // weight-gradient co tiles per wave
};

// CIN -> COUT = 128 -> 128 (consumer wave w: data-gradient ci tile w, weight-gradient tiles [all 4 co][ci tile w]) or 64 -> 128
// (waves 0,1: data-gradient ci tile w; wave w: weight-gradient tiles [co tiles 2 (w >> 1), +1][ci tile w & 1]; the fp32-MFMA
// form of this layer was matrix-bound).
template <int CIN, int COUT>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
pw_bwd_fused16_k(const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ bn_pw, const uint16_t* __restrict__ wq,
                 const float* __restrict__ wraw, const float* __restrict__ wmax, const float* __restrict__ ydw, const float* __restrict__ bn_dw, float* __restrict__ g_dw,
                 float* __restrict__ dW, float* __restrict__ wpartial, float* __restrict__ part, int64_t M, int ntiles) {
  using S_ = F16Shape<CIN, COUT>;
  constexpr int BM = kF16BM, NCI = S_::NCI, COW = S_::COW, kStage = S_::Stage, kRowPlane = S_::RowPlane, kTY = S_::TPlaneY, kTA = S_::TPlaneA,
                kLdc = S_::Ldc;
  static_assert((CIN == 128 || CIN == 64) && COUT == 128, "shapes");
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * kStage + 4 * 2 * 32 * 4];
  float* red = reinterpret_cast<float*>(lds + 2 * kStage);  // [4][2][32]: the data-gradient waves' 32 channels each
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool producer = wave >= 4;
  const int r32 = lane & 31, h = lane >> 5;
  const float sa = pow2_scale(bn_pw[(size_t)TTK_BN_AUX * COUT + TTK_AUX_DY_BOUND]);
  const float sx = pow2_scale(bn_dw[(size_t)TTK_BN_AUX * CIN + TTK_AUX_ACT_BOUND]);
  const float sw = pow2_scale(*wmax);
  auto stage_ptr = [&](int st) { return lds + st * kStage; };

  if (producer) {
    const int pt = tid - 256;
    const int mb = pt & 7, cq = pt >> 3;  // 8 row blocks of 4 rows x 32 channel quads (COUT = 128; the first CIN / 4 of them also carry ydw)
    const bool a_on = cq < CIN / 4;
    const int cqa = a_on ? cq : 0;
    const int sub = mb >> 2, chunk = (mb >> 1) & 1, o8 = (mb & 1) * 8;
    const fz32x4 ga = *reinterpret_cast<const fz32x4*>(bn_pw + TTK_BN_GA * COUT + 4 * cq) * sa;
    const fz32x4 gb = *reinterpret_cast<const fz32x4*>(bn_pw + TTK_BN_GB * COUT + 4 * cq) * sa;
    const fz32x4 gmean = *reinterpret_cast<const fz32x4*>(bn_pw + TTK_BN_GMEAN * COUT + 4 * cq);
    const fz32x4 ymean = *reinterpret_cast<const fz32x4*>(bn_pw + TTK_BN_MEAN * COUT + 4 * cq);
    const fz32x4 dsc = *reinterpret_cast<const fz32x4*>(bn_dw + TTK_BN_SCALE * CIN + 4 * cqa) * sx;
    const fz32x4 dbe = *reinterpret_cast<const fz32x4*>(bn_dw + TTK_BN_BETA * CIN + 4 * cqa) * sx;
    const fz32x4 dmean = *reinterpret_cast<const fz32x4*>(bn_dw + TTK_BN_MEAN * CIN + 4 * cqa);
    fz32x4 rg[4], ry[4], rc[4];
    auto load = [&](int t) {
      const int64_t r0 = (int64_t)t * BM + 4 * mb;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int64_t row = r0 + i < M ? r0 + i : M - 1;
        const float4 a = ld4nt(g + act_off(row, 4 * cq, M)), b = ld4nt(y + act_off(row, 4 * cq, M));  // activations: channel blocks (ttk_common.h)
        rg[i] = fz32x4{a.x, a.y, a.z, a.w}; ry[i] = fz32x4{b.x, b.y, b.z, b.w};
        if (a_on) {
          const float4 c = ld4(ydw + act_off(row, 4 * cqa, M));
          rc[i] = fz32x4{c.x, c.y, c.z, c.w};
        }
      }
    };
    auto store = [&](int t, int st) {
      unsigned char* S = stage_ptr(st);
      unsigned char* DyR = S;
      unsigned char* DyT = S + 2 * kRowPlane;
      unsigned char* AT = DyT + 2 * 2 * kTY;
      float* Yc = reinterpret_cast<float*>(AT + 2 * 2 * kTA);
      const int64_t r0 = (int64_t)t * BM + 4 * mb;
      fz32x4 dy[4], av[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dy[i] = ga * (rg[i] - gmean) + gb * (ry[i] - ymean);
        if (r0 + i >= M) dy[i] = fz32x4{0.f, 0.f, 0.f, 0.f};  // rows past the end contribute nothing
        // dy row-major: 8-byte piece cq of row 4 mb + i, 16-byte chunk XOR (row & 15)
        uint2 ph, pl;
        fz_split(dy[i], ph, pl);
        const int row = 4 * mb + i, off = row * (COUT * 2) + ((((cq >> 1) ^ (row & 15))) << 4) + (cq & 1) * 8;
        *reinterpret_cast<uint2*>(DyR + off) = ph;
        *reinterpret_cast<uint2*>(DyR + kRowPlane + off) = pl;
        if (a_on) {
          const fz32x4 yc = rc[i] - dmean;
          fz32x4 a = dsc * yc + dbe;
          a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
          if (r0 + i >= M) a = fz32x4{0.f, 0.f, 0.f, 0.f};
          av[i] = a;
          *reinterpret_cast<fz32x4*>(Yc + (4 * mb + i) * kLdc + 4 * cqa) = yc;
        }
      }
      // transposed: channel 4 cq + e, four consecutive rows 4 mb .. 4 mb + 3 (k) -> 8 bytes per plane
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint2 ph, pl;
        fz_split(fz32x4{dy[0][e], dy[1][e], dy[2][e], dy[3][e]}, ph, pl);
        const int offy = sub * (2 * kTY) + fz_swz(4 * cq + e, chunk) + o8;
        *reinterpret_cast<uint2*>(DyT + offy) = ph;
        *reinterpret_cast<uint2*>(DyT + kTY + offy) = pl;
        if (a_on) {
          fz_split(fz32x4{av[0][e], av[1][e], av[2][e], av[3][e]}, ph, pl);
          const int offa = sub * (2 * kTA) + fz_swz(4 * cqa + e, chunk) + o8;
          *reinterpret_cast<uint2*>(AT + offa) = ph;
          *reinterpret_cast<uint2*>(AT + kTA + offa) = pl;
        }
      }
    };
    int t = blockIdx.x;
    if (t < ntiles) {
      load(t);
      store(t, 0);
      if (t + (int)gridDim.x < ntiles) load(t + gridDim.x);
    }
    __syncthreads();
    for (int it = 0; t < ntiles; t += gridDim.x, ++it) {
      const int tn = t + gridDim.x;
      if (tn < ntiles) {
        store(tn, (it + 1) & 1);
        if (tn + (int)gridDim.x < ntiles) load(tn + gridDim.x);
      }
      __syncthreads();
    }
  } else {
    const bool has_dg = wave < NCI;                 // data gradient: ci tile `wave`
    const int cit = wave % NCI, cog = wave / NCI;   // weight gradient: ci tile, group of co tiles
    const int ci_dg = 32 * (has_dg ? wave : 0) + r32, ci_wg = 32 * cit + r32;
    // data-gradient B fragments: W^T planes [co / 32][ci][32] of the prepared operand, k16 step s, lane half h: co 16 s + 8 h ..
    fz16x8 Wf[COUT / 16][2];
#pragma unroll
    for (int s = 0; s < COUT / 16; ++s) {
      const int co = 16 * s + 8 * h;
      if (CIN == 128) {
        const size_t idx = ((size_t)(co >> 5) * CIN + ci_dg) * 32 + (co & 31);
        Wf[s][0] = *reinterpret_cast<const fz16x8*>(wq + idx);
        Wf[s][1] = *reinterpret_cast<const fz16x8*>(wq + (size_t)CIN * COUT + idx);
      } else {
        // 64 -> 128: the prepared block of this layer holds no fp16 data-gradient planes (its stand-alone data gradient runs on
        // fp32 MFMA) - cut the raw weights here, once per workgroup, with the block's |w| scale
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float xs = wraw[(size_t)(co + e) * CIN + ci_dg] * sw;
          const _Float16 hh = (_Float16)xs;
          Wf[s][0][e] = hh;
          Wf[s][1][e] = (_Float16)(xs - (float)hh);
        }
      }
    }
    const float dsc = bn_dw[TTK_BN_SCALE * CIN + ci_dg], dbe = bn_dw[TTK_BN_BETA * CIN + ci_dg];
    const float inv_dg = 1.f / (sa * sw), inv_wg = 1.f / (sa * sx);
    f32x16 wacc[COW];
#pragma unroll
    for (int tt = 0; tt < COW; ++tt)
#pragma unroll
      for (int e = 0; e < 16; ++e) wacc[tt][e] = 0.f;
    float s1 = 0.f, s2 = 0.f;
    __syncthreads();  // stage 0
    int it = 0;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x, ++it) {
      const unsigned char* S = stage_ptr(it & 1);
      const unsigned char* DyR = S;
      const unsigned char* DyT = S + 2 * kRowPlane;
      const unsigned char* AT = DyT + 2 * 2 * kTY;
      const float* Yc = reinterpret_cast<const float*>(AT + 2 * 2 * kTA);
      const int64_t m0 = (int64_t)t * BM;
      // ---- data gradient: 32 rows x this wave's 32 ci, contraction over the co
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      if (has_dg) {
#pragma unroll
        for (int s = 0; s < COUT / 16; ++s) {
          const int off = r32 * (COUT * 2) + ((((2 * s + h) ^ (r32 & 15))) << 4);
          const fz16x8 ah = *reinterpret_cast<const fz16x8*>(DyR + off), al = *reinterpret_cast<const fz16x8*>(DyR + kRowPlane + off);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Wf[s][1], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Wf[s][0], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Wf[s][0], acc, 0, 0, 0);
        }
      }
      // ---- weight gradient: dW[co tile][this wave's ci tile] += dy^T a over the 32 rows (two k16 stages)
      fz16x8 bh[2], bl[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int off = s * (2 * kTA) + fz_swz(ci_wg, h);
        bh[s] = *reinterpret_cast<const fz16x8*>(AT + off);
        bl[s] = *reinterpret_cast<const fz16x8*>(AT + kTA + off);
      }
#pragma unroll
      for (int tt = 0; tt < COW; ++tt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int off = s * (2 * kTY) + fz_swz(32 * (cog * COW + tt) + r32, h);
          const fz16x8 ah = *reinterpret_cast<const fz16x8*>(DyT + off), al = *reinterpret_cast<const fz16x8*>(DyT + kTY + off);
          wacc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[s], wacc[tt], 0, 0, 0);
          wacc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[s], wacc[tt], 0, 0, 0);
          wacc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[s], wacc[tt], 0, 0, 0);
        }
      // ---- data-gradient epilogue: accumulator element e of lane (r32, h) = row (e & 3) + 8 (e >> 2) + 4 h, column ci
      if (has_dg) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
          const float yc = Yc[row * kLdc + ci_dg];
          const float out = fmaf(dsc, yc, dbe) > 0.f ? acc[e] * inv_dg : 0.f;
          if (m0 + row < M) {
            g_dw[act_off(m0 + row, ci_dg, M)] = out;
            s1 += out;
            s2 = fmaf(out, yc, s2);
          }
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int tt = 0; tt < COW; ++tt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = 32 * (cog * COW + tt) + (e & 3) + 8 * (e >> 2) + 4 * h;
        const float v = wacc[tt][e] * inv_wg;
        if (wpartial) wpartial[(size_t)blockIdx.x * COUT * CIN + co * CIN + ci_wg] = v;
        else atomicAdd(dW + co * CIN + ci_wg, v);
      }
    s1 += __shfl_xor(s1, 32);
    s2 += __shfl_xor(s2, 32);
    if (h == 0 && has_dg) { red[(wave * 2 + 0) * 32 + r32] = s1; red[(wave * 2 + 1) * 32 + r32] = s2; }
  }
  __syncthreads();
  if (part && tid < 2 * CIN) {
    const int which = tid / CIN, c = tid % CIN;
    part[(size_t)blockIdx.x * 2 * CIN + which * CIN + c] = red[((c >> 5) * 2 + which) * 32 + (c & 31)];
  }
}

// 128 -> 128 always runs on the fp16 pipe; 64 -> 128 too when the prepared block holds fp16 planes (default TTK_GEMM mode), else
// on fp32 MFMA from the raw weights
static bool fused_f16(int Cin) { return Cin == 128 || (Cin == 64 && gemm_mode() == GEMM_F16X2 && !exp_env("TTK_FUSED_FP32")); }
static bool fused_shape(int Cin, int Cout) {
  // (the 128 -> 128 form reads the fp16 planes of the prepared weight block: default TTK_GEMM mode only)
  return (Cin == 32 && Cout == 64) || (Cin == 64 && Cout == 128) || (Cin == 128 && Cout == 128 && gemm_mode() == GEMM_F16X2);
}

static int fused_grid(int64_t M, int Cin) {
  const int64_t ntiles = ceil_div(M, fused_f16(Cin) ? kF16BM : kFusedBM);
  const int64_t cap = Cin == 32 ? 512 : 256;  // two workgroups per CU fit for the 32 -> 64 layer (62 KB of LDS), one for the others
  return (int)(ntiles < cap ? ntiles : cap);
}

}  // namespace ttk

using namespace ttk;

extern "C" {

int ttk_pwconv1x1_bwd_fused_rows(int64_t M, int Cin, int Cout) { return fused_shape(Cin, Cout) && M > 0 ? fused_grid(M, Cin) : 0; }

size_t ttk_pwconv1x1_bwd_fused_partial_bytes(int64_t M, int Cin, int Cout) {
  return fused_shape(Cin, Cout) && M > 0 ? (size_t)fused_grid(M, Cin) * Cin * Cout * sizeof(float) : 0;
}

int ttk_pwconv1x1_bwd_fused(const float* g, const float* y, const float* bn_pw, const float* w, const void* wsplit, const float* ydw,
                            const float* bn_dw, float* g_dw, float* dw, float* partial, float* part, int64_t M, int Cin, int Cout,
                            ttk_stream_t stream) {
  TTK_REQUIRE(g && y && bn_pw && ydw && bn_dw && g_dw && dw, "pwconv1x1_bwd_fused: null pointer");
  TTK_REQUIRE(fused_shape(Cin, Cout) && M > 0, "pwconv1x1_bwd_fused: only the 32 -> 64, 64 -> 128 and 128 -> 128 layers (got %d -> %d)", Cin, Cout);
  TTK_REQUIRE((Cin == 128 || w) && (!fused_f16(Cin) || wsplit),
              "pwconv1x1_bwd_fused: the fp16 forms (128 -> 128; 64 -> 128 in the default mode) need the prepared weight block, 32 -> 64 and 64 -> 128 the raw weights");
  const int ntiles = (int)ceil_div(M, fused_f16(Cin) ? kF16BM : kFusedBM), grid = fused_grid(M, Cin);
  hipStream_t st = (hipStream_t)stream;
  if (fused_f16(Cin)) {
    // prepared block (ttk_pwconv_prepare_weights, fp16 mode): [forward planes 4n][data-gradient planes 4n][|w| maximum]
    const size_t n = (size_t)Cin * Cout;
    const unsigned char* ws = static_cast<const unsigned char*>(wsplit);
    const uint16_t* wq = reinterpret_cast<const uint16_t*>(ws + prep_bwd_offset(n));
    const float* wmx = reinterpret_cast<const float*>(ws + prep_hdr_offset(n));
    if (Cin == 128)
      hipLaunchKernelGGL((pw_bwd_fused16_k<128, 128>), dim3(grid), dim3(512), 0, st, g, y, bn_pw, wq, w, wmx, ydw, bn_dw, g_dw, dw, partial, part, M, ntiles);
    else
      hipLaunchKernelGGL((pw_bwd_fused16_k<64, 128>), dim3(grid), dim3(512), 0, st, g, y, bn_pw, wq, w, wmx, ydw, bn_dw, g_dw, dw, partial, part, M, ntiles);
  } else if (Cin == 32) {
    hipLaunchKernelGGL((pw_bwd_fused_k<32, 64>), dim3(grid), dim3(512), 0, st, g, y, bn_pw, w, ydw, bn_dw, g_dw, dw, partial, part, M, ntiles);
  } else {
    hipLaunchKernelGGL((pw_bwd_fused_k<64, 128>), dim3(grid), dim3(512), 0, st, g, y, bn_pw, w, ydw, bn_dw, g_dw, dw, partial, part, M, ntiles);
  }
  if (partial) launch_fold_partials(partial, grid, (int64_t)Cin * Cout, dw, 1, st);
  TTK_LAUNCH_CHECK("pwconv1x1_bwd_fused");
}

}  // extern "C"
