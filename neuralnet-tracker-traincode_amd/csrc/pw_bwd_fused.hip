// Backward of the first pointwise convolutions (32 -> 64 and 64 -> 128 channels; DepthWiseBlock.conv_sep + bn_sep,
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
        rg[i] = ld4nt(g + rcl * COUT + 4 * cy);
        ry[i] = ld4nt(y + rcl * COUT + 4 * cy);
      }
#pragma unroll
      for (int i = 0; i < NC; ++i) {
        const int64_t row = m0 + (pt + 256 * i) / QC;
        const int64_t rcl = row < M ? row : M - 1;
        rc[i] = ld4(ydw + rcl * CIN + 4 * cc);  // (read again by the depthwise backward: cached)
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
              g_dw[(size_t)(m0 + row) * CIN + col] = out;  // 32 lanes = 128 contiguous bytes
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

static bool fused_shape(int Cin, int Cout) { return (Cin == 32 && Cout == 64) || (Cin == 64 && Cout == 128); }

static int fused_grid(int64_t M, int Cin) {
  const int64_t ntiles = ceil_div(M, kFusedBM);
  const int64_t cap = Cin == 32 ? 512 : 256;  // two workgroups per CU fit for the 32 -> 64 layer (62 KB of LDS), one for 64 -> 128
  return (int)(ntiles < cap ? ntiles : cap);
}

}  // namespace ttk

using namespace ttk;

extern "C" {

int ttk_pwconv1x1_bwd_fused_rows(int64_t M, int Cin, int Cout) { return fused_shape(Cin, Cout) && M > 0 ? fused_grid(M, Cin) : 0; }

size_t ttk_pwconv1x1_bwd_fused_partial_bytes(int64_t M, int Cin, int Cout) {
  return fused_shape(Cin, Cout) && M > 0 ? (size_t)fused_grid(M, Cin) * Cin * Cout * sizeof(float) : 0;
}

int ttk_pwconv1x1_bwd_fused(const float* g, const float* y, const float* bn_pw, const float* w, const float* ydw, const float* bn_dw,
                            float* g_dw, float* dw, float* partial, float* part, int64_t M, int Cin, int Cout, ttk_stream_t stream) {
  TTK_REQUIRE(g && y && bn_pw && w && ydw && bn_dw && g_dw && dw, "pwconv1x1_bwd_fused: null pointer");
  TTK_REQUIRE(fused_shape(Cin, Cout) && M > 0, "pwconv1x1_bwd_fused: only the 32 -> 64 and 64 -> 128 layers (got %d -> %d)", Cin, Cout);
  const int ntiles = (int)ceil_div(M, kFusedBM), grid = fused_grid(M, Cin);
  hipStream_t st = (hipStream_t)stream;
  if (Cin == 32) {
    hipLaunchKernelGGL((pw_bwd_fused_k<32, 64>), dim3(grid), dim3(512), 0, st, g, y, bn_pw, w, ydw, bn_dw, g_dw, dw, partial, part, M, ntiles);
  } else {
    hipLaunchKernelGGL((pw_bwd_fused_k<64, 128>), dim3(grid), dim3(512), 0, st, g, y, bn_pw, w, ydw, bn_dw, g_dw, dw, partial, part, M, ntiles);
  }
  if (partial) launch_fold_partials(partial, grid, (int64_t)Cin * Cout, dw, 1, st);
  TTK_LAUNCH_CHECK("pwconv1x1_bwd_fused");
}

}  // extern "C"
