// Pointwise 1x1 convolutions of the wide layers (Cin >= 128, Cout a multiple of 256) - forward and data gradient - as
// fp32 GEMMs on the fp16 matrix pipe, "row-block" form.  Reference: DepthWiseBlock.conv_sep + bn_sep,
// backbones/mobilenet_v1.py:67-68,82-84.  Arithmetic, operand bounds and numerics: pwconv_f16.hip (two fp16 pieces per
// operand, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulation).
//
// What is different from pw16_k, and why (round 3; measurements in DESIGN.md 4.1).  A 128x256 tile moves 16 KB of A (fp32
// rows from HBM, ~12 B/clk per CU when every CU streams) and 32 KB of weight planes (L2, ~35 B/clk) per k32 step through
// its CU for 1536 cycles of matrix work per SIMD - the bytes, not the MFMAs, set the step - and M = 41 472 rows make 648
// tiles: 2.53 rounds on 256 CUs, paid as 3.  Here
//  * a workgroup is 12 waves: EIGHT consumer waves (two per SIMD: one wave's fragment reads hide behind the other's MFMAs)
//    on a tile of up to 256 rows x 256 columns, four producer waves;
//  * the tile HEIGHT is not fixed: the host cuts M into row blocks of RT <= 32 RBLK rows such that the tiles fill whole
//    rounds of the 256 CUs (41 472 x 512: 256 row blocks of 162 rows x 2 column tiles = exactly 2 rounds; the MFMAs run
//    on the padded 192 rows, which the byte-bound step has room for);
//  * the weight planes reach LDS by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write), issued by the CONSUMER waves right after
//    the barrier that frees a ring slot (four 1 KB pieces per wave and step: they have the whole step to land; issued by the producers
//    behind their A work they had half a step and the kernel was no faster than pw16_k): ttk_pwconv_prepare_weights
//    stores them as [K/16][Nout][16] with the consumers' chunk swizzle already applied, so a k16 stage of 256 rows is ONE
//    contiguous 8 KB block per piece plane and a wave-instruction copies 1 KB of it;
//  * BatchNorm partial sums: one row per tile (ttk_partial_rows_pwconv).
#include "ttk_common.h"
#include "conv_geom.h"
#include <type_traits>

// TTK_M_NOPK=1 (experiment builds): this file's kernels without packed fp32 instructions.  Measured (profiles/r04_rowblock_gemm_variants.txt):
// pw16m_k 5 % SLOWER (forward 87.8 vs 82.9 us, data gradient 103 vs 96) - beside MFMAs the NUMBER of vector instructions is what costs.
#ifndef TTK_M_NOPK
#define TTK_M_NOPK 0
#endif
#if TTK_M_NOPK && defined(__HIP_DEVICE_COMPILE__)
// every function of this translation unit (kernels, their lambdas, the inline helpers): same target features, so everything still inlines
#pragma clang attribute push(__attribute__((target("no-packed-fp32-ops"))), apply_to = function)
#endif

namespace ttk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

enum { RMODE_FWD = 0, RMODE_DGRAD = 1 };
// Experiment builds only (tools/exp/build_variants.sh; the product is built with all of them at their defaults):
//   TTK_R_DBG   timing-only bits, results wrong by construction: 1 no A loads | 2 no LDS-DMA | 4 no MFMAs | 8 no epilogue |
//               16 the data gradient's epilogue does not read its mask operand | 32 the producers store their rows unconverted |
//               64 the producers store nothing (barriers only) | 128 the consumers read no fragments (MFMAs on stale registers)
//   TTK_R_PPRIO / TTK_R_CPRIO   s_setprio of the producer / consumer waves
#ifndef TTK_R_DBG
#define TTK_R_DBG 0
#endif
#ifndef TTK_R_PPRIO
#define TTK_R_PPRIO 3
#endif
#ifndef TTK_R_CPRIO
#define TTK_R_CPRIO 0
#endif
//   TTK_R_SETS  register sets of the producers' A rows: 2 (default) = the rows of step s + 2 are requested before step s + 1 is
//               converted; 1 = round 3's schedule (rows requested one step ahead, after the conversion)
#ifndef TTK_R_SETS
#define TTK_R_SETS 2
#endif
//   TTK_R_GPS_F / TTK_R_NSLOT_F, TTK_R_GPS_D / TTK_R_NSLOT_D   groups per step and register slots of the producers' pipeline
//               (forward / data gradient; see pw16r_k)
//   TTK_R_MERGED 1 (default) = pw16m_k (eight waves that convert and multiply) for the data gradient, 2 = for the forward too,
//                0 = pw16r_k (eight MFMA + four producer waves) everywhere
//   TTK_M_FENCE  1 = a scheduling fence behind every conversion part of pw16m_k
#ifndef TTK_R_MERGED
#define TTK_R_MERGED 1
#endif
#ifndef TTK_M_FENCE
#define TTK_M_FENCE 1
#endif
//   TTK_M_MIX    1 (default) = low pieces by v_fma_mix, 0 = convert back, subtract, convert
#ifndef TTK_M_MIX
#define TTK_M_MIX 1
#endif

#ifndef TTK_R_GPS_F
#define TTK_R_GPS_F 1
#endif
#ifndef TTK_R_NSLOT_F
#define TTK_R_NSLOT_F 2
#endif
#ifndef TTK_R_GPS_D
#define TTK_R_GPS_D 2
#endif
#ifndef TTK_R_NSLOT_D
#define TTK_R_NSLOT_D 3
#endif
template <int N, typename F, int I = 0>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, F, I + 1>(static_cast<F&&>(f));
  }
}
constexpr int kRDbg = TTK_R_DBG;
__device__ __forceinline__ f32x16 rmfma(f16x8 a, f16x8 b, f32x16 c) {
  if constexpr (kRDbg & 4) {
    asm volatile("" ::"v"(a), "v"(b));
    return c;
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
}
constexpr int kRBN = 256;  // tile width

// raw s_barrier (no vmcnt drain: the producers keep global loads in flight across it) fenced against compiler motion of LDS accesses
__device__ __forceinline__ void rbarrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

#if defined(TTK_R_STAMP)
// Experiment builds only (tools/exp): cycle accounting per wave.  A stamp is s_memtime behind the lgkmcnt(0) a barrier needs anyway.
__device__ unsigned long long g_r_stamps[2048 * 12 * 8];
#define TTK_STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define TTK_RSTAMP(t) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
__device__ __forceinline__ void rbarrier_s(unsigned long long& wait) {
  unsigned long long a, b;
  TTK_STAMP(a);
  __builtin_amdgcn_s_barrier();
  TTK_STAMP(b);
  wait += b - a;
}
#endif

__device__ __forceinline__ int rswz(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 3) & 1)) << 4); }

__device__ __forceinline__ void rsplit_store(f32x4 v, unsigned char* dst, int plane) {
  const f16x2 h01 = __builtin_convertvector(f32x2{v.x, v.y}, f16x2), h23 = __builtin_convertvector(f32x2{v.z, v.w}, f16x2);
  const f32x2 f01 = __builtin_convertvector(h01, f32x2), f23 = __builtin_convertvector(h23, f32x2);
  const f16x2 l01 = __builtin_convertvector(f32x2{v.x - f01.x, v.y - f01.y}, f16x2);
  const f16x2 l23 = __builtin_convertvector(f32x2{v.z - f23.x, v.w - f23.y}, f16x2);
  *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
  *reinterpret_cast<uint2*>(dst + plane) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}

// Low piece l = fp16(x - h) of two values whose high pieces are the halves of `h`: v_fma_mixlo/hi_f16 take the fp16 half as a source
// of an fp32 fma and round the result to fp16 once - the value of (cvt_f32_f16, subtract, cvt_pk_f16_f32) in two instructions
// instead of three and a half.  (Vector instructions of an MFMA wave are not free: they cost the matrix pipe a few cycles each.)
__device__ __forceinline__ unsigned rlow2(unsigned h, float x0, float x1) {
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(l)
      : "v"(h), "v"(x0), "v"(x1));
  return l;
}
template <typename T>
__device__ __forceinline__ f32x4 rld_act4(const T* p) {
  const float4 v = Act<T>::ldnt(p);
  return f32x4{v.x, v.y, v.z, v.w};
}
// The producers' row loads as BUFFER loads: a wave-uniform descriptor (rebuilt per k32 step: scalar work) + one 32-bit byte offset
// per lane.  As global loads hipcc kept a 64-bit address pair per (row pass, tensor) in registers across the loop - 24 VGPRs of the
// data gradient's producers, which is what made a second set of rows spill.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rbuf(const T* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, 0x7fffffff, 0x00020000);
}
template <typename T>
__device__ __forceinline__ f32x4 rbuf_ld4(__amdgpu_buffer_rsrc_t r, unsigned elem_off) {  // 4 consecutive elements, streaming (nt)
  if constexpr (Act<T>::kBf16) {
    const u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, elem_off * 2u, 0, 2);
    const float4 v = Act<bf16_t>::widen(make_uint2(u.x, u.y));
    return f32x4{v.x, v.y, v.z, v.w};
  } else {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, elem_off * 4u, 0, 2));
  }
}

// Geometry of a tile of RBLK 32-row blocks x 8 32-column blocks on 8 consumer waves
template <int RBLK>
struct RGeo {
  static constexpr int RB = 32 * RBLK;
  static constexpr int WM = RBLK == 8 ? 4 : 2, WN = 8 / WM;  // consumer waves along M / N
  static constexpr int TM = RBLK / WM, TN = 8 / WN;         // 32 x 32 blocks per consumer wave
  static constexpr int APL = RB * 32, BPL = kRBN * 32;      // bytes of one piece plane of a k16 stage
  static constexpr int kStage = 2 * APL + 2 * BPL;          // [A h][A l][B h][B l]
  static constexpr int kStr = kStage + 64;                  // the two k16 halves of a producer's ds_write_b64 use different banks
  static constexpr int kRing = 2 * 2 * kStr;                // two k32 super-stages
  static constexpr int CH = 32 * WM;                        // rows of one epilogue chunk (block i of every consumer wave)
  static constexpr int LDC = kRBN + 4;
  static constexpr int kEpi = CH * LDC * 4 + 12 * 2 * kRBN * 4;
  static constexpr int kCst = 4 * 1024 * 4;                 // per-channel constants of the A operand: [4][K <= 1024] floats behind the ring (main loop only)
  static constexpr int kSmem = kRing + kCst > kEpi ? kRing + kCst : kEpi;
};

// ---------------- epilogue of a tile, all NW waves: the tile leaves through LDS in TM chunks of CH rows (block c of every MFMA wave) -
// un-scale, 16-byte stores (a wave writes 1 KB row segments), the data gradient's ReLU mask, BatchNorm sums ----------------
template <int RBLK, int MODE, int NW, typename T, typename TO, typename Acc, bool M16 = false>
__device__ __forceinline__ void r_epilogue(unsigned char* lds, Acc& acc, TO* __restrict__ out, const T* __restrict__ E0, const float* __restrict__ bnE,
                                           float* __restrict__ part, int64_t M, int Nout, int64_t m0, int64_t m_end, int n0, unsigned by, float sa, float sb) {
  using G = RGeo<RBLK>;
  constexpr int WM = G::WM, WN = G::WN, TM = G::TM, TN = G::TN;
  constexpr bool FWD = MODE == RMODE_FWD;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int LDC = G::LDC, CH = G::CH;
  float* Cs = reinterpret_cast<float*>(lds);
  float* red = reinterpret_cast<float*>(lds + CH * LDC * 4);  // [NW][2][256]
  // 2 NW half-waves = 8 channel blocks x NW / 4 row phases; a half-wave = 4 consecutive rows x the 8 quads of ONE channel block: 512
  // contiguous bytes of the output (and of the mask operand) per half-wave and instruction in the channel-block layout.  The column
  // quad of a thread is fixed (its partial sums), its rows are rg, rg + NW, ...
  const int hw_ = tid >> 5, c4 = (hw_ & 7) * 8 + (tid & 7), rg = (hw_ >> 3) * 4 + ((tid >> 3) & 3);  // NW row groups
  const int col = n0 + 4 * c4;
  const float inv = 1.f / (sa * sb);  // exact: a power of two
  float4 esc = f4(0.f), emean = f4(0.f), ebeta = f4(0.f);
  if constexpr (!FWD) {
    esc = ld4(bnE + TTK_BN_SCALE * Nout + col); emean = ld4(bnE + TTK_BN_MEAN * Nout + col); ebeta = ld4(bnE + TTK_BN_BETA * Nout + col);
  } else {
    if (bnE) emean = ld4(bnE + col);  // forward: bnE is the statistics pivot [Nout] (ttk.h), the sums are those of y - pivot
  }
  float4 s1 = f4(0.f), s2 = f4(0.f);
  if constexpr (kRDbg & 8) {
    if (wave < 8) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
    }
  } else {
  // The data gradient's mask operand (raw depthwise output) is requested one group of rows ahead into the other of two register
  // buffers: a chunk's later groups and the next chunk's first group while the current group is processed (the registers of the
  // accumulators already parked are free by then), the tile's first group right after chunk 0 is parked.  Requested inside the
  // row loop, two rows at a time, every round trip was exposed: 23 000 cycles per tile against the forward's 9 600
  // (profiles/r04_rowblock_stamps.txt).
  constexpr int NRG = NW == 12 ? 6 : 8, NGRP = (CH + NW * NRG - 1) / (NW * NRG);  // rows per thread and group; groups per chunk
  float4 ev[2][NRG];
  auto tile_row = [&](int c, int cr) -> int64_t { return m0 + (cr >> 5) * (32 * TM) + c * 32 + (cr & 31); };  // chunk row -> tile row: block c of consumer row cr / 32
  auto e_load = [&](auto idxc) {  // group idx = c * NGRP + g -> buffer idx & 1
    constexpr int idx = decltype(idxc)::value, c = idx / NGRP, g = idx % NGRP;
    if constexpr (!FWD && idx < TM * NGRP) {
#pragma unroll
      for (int u = 0; u < NRG; ++u) {
        const int cr = rg + NW * (g * NRG + u);
        const int64_t grow = tile_row(c, cr);
        ev[idx & 1][u] = f4(0.f);
        if constexpr (!(kRDbg & 16))
          if (cr < CH && grow < m_end) ev[idx & 1][u] = Act<T>::ld(E0 + act_off(grow, col, M));
      }
    }
  };
  static_for<TM>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if (wave < 8) {
      const int lane = tid & 63, wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
      if constexpr (M16) {  // acc[2 TM][2 TN] of 16 x 16 blocks (four registers: row 4 (lane >> 4) + i, column lane & 15)
        const int r16 = lane & 15, g4 = lane >> 4;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
          for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) Cs[(wm * 32 + s_ * 16 + 4 * g4 + i) * LDC + wn * (32 * TN) + j * 16 + r16] = acc[2 * c + s_][j][i];
      } else {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) Cs[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * LDC + wn * (32 * TN) + j * 32 + r] = acc[c][j][e];
      }
    }
    if constexpr (c == 0) e_load(std::integral_constant<int, 0>{});
    __syncthreads();  // chunk c is in LDS
    static_for<NGRP>([&](auto gc) {
      constexpr int g = decltype(gc)::value, idx = c * NGRP + g;
      e_load(std::integral_constant<int, idx + 1>{});
#pragma unroll
      for (int u = 0; u < NRG; ++u) {
        const int cr = rg + NW * (g * NRG + u);
        const int64_t grow = tile_row(c, cr);
        if (cr >= CH || grow >= m_end) continue;
        float4 v = ld4(Cs + cr * LDC + 4 * c4);
        v = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
        const size_t o = act_off(grow, col, M);
        if constexpr (FWD) {
          v = Act<TO>::round(v);  // statistics of what is stored
          Act<TO>::st(out + o, v);
          v = sub4(v, emean);
          s1 = add4(s1, v);
          s2 = fma4(v, v, s2);
        } else {
          const float4 yc = sub4(ev[idx & 1][u], emean);
          v = Act<TO>::round(mask4(v, fma4(esc, yc, ebeta)));
          Act<TO>::st(out + o, v);
          s1 = add4(s1, v);
          s2 = fma4(v, yc, s2);
        }
      }
    });
    __syncthreads();  // the row pass is done: the next chunk may be parked
  });
  }
  if (part) {  // one row of partial sums per tile: NW row groups folded in a fixed order
    st4(red + (rg * 2 + 0) * kRBN + 4 * c4, s1);
    st4(red + (rg * 2 + 1) * kRBN + 4 * c4, s2);
    __syncthreads();
    if (tid < 2 * kRBN) {
      const int which = tid / kRBN, c = tid % kRBN;
      float a = 0.f;
#pragma unroll
      for (int qq = 0; qq < NW; ++qq) a += red[(qq * 2 + which) * kRBN + c];
      part[(size_t)by * 2 * Nout + (size_t)which * Nout + n0 + c] = a;
    }
  }
}

// A: fp32 rows [M][K], formed on load (forward: relu(bn(y)); data gradient: ga*(g-gmean)+gb*(y-mean)); Bq: two fp16 planes
// [K/16][Nout][16] (chunk-swizzled) of the weights scaled by pow2_scale(*wmax).  Tile t: rows [by*RT, min((by+1)*RT, M)),
// columns [bx*256, +256); part[by][2][Nout].
template <int RBLK, int MODE, typename T, typename TO>
__global__ void __launch_bounds__(768) pw16r_k(const TO* __restrict__ A0, const T* __restrict__ A1, const float* __restrict__ bnA,
                                               const uint16_t* __restrict__ Bq, const float* __restrict__ wmax, TO* __restrict__ out,
                                               const T* __restrict__ E0, const float* __restrict__ bnE, float* __restrict__ part, int64_t M,
                                               int K, int Nout, int RT) {
  using G = RGeo<RBLK>;
  constexpr int RB = G::RB, WM = G::WM, WN = G::WN, TM = G::TM, TN = G::TN, APL = G::APL, BPL = G::BPL, kStr = G::kStr;
  constexpr bool FWD = MODE == RMODE_FWD;
  __shared__ __attribute__((aligned(16))) unsigned char lds[G::kSmem];

  const int tid = threadIdx.x;
  // XCD-aware tile order: every XCD gets a contiguous range of tiles, so the column tiles of one row block (same A rows) run
  // side by side on one L2
  const unsigned Gd = gridDim.x, Lid = blockIdx.x, NB = Nout / kRBN;
  const unsigned xq = Gd / 8, xr = Gd % 8, xcd = Lid % 8;
  const unsigned tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + Lid / 8;
  const unsigned bx = tile % NB, by = tile / NB;
  const int64_t m0 = (int64_t)by * RT;
  const int64_t m_end = m0 + RT < M ? m0 + RT : M;
  const int n0 = bx * kRBN;
  const int nks = K / 32;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float sa = pow2_scale(bnA[(size_t)TTK_BN_AUX * K + (FWD ? TTK_AUX_ACT_BOUND : TTK_AUX_DY_BOUND)]);
  const float sb = pow2_scale(*wmax);
  f32x16 acc[TM][TN];  // consumer waves only
  // per-channel constants of the A operand, once per tile: [scale S_a | mean | beta S_a] (forward), [ga S_a | gmean | gb S_a | mean] (data gradient)
  float* cst = reinterpret_cast<float*>(lds + G::kRing);
  {
    constexpr int NQ = FWD ? 3 : 4;
    for (int i = tid * 4; i < NQ * K; i += 768 * 4) {
      const int j = i / K, c = i - j * K;
      const int row = FWD ? (j == 0 ? TTK_BN_SCALE : (j == 1 ? TTK_BN_MEAN : TTK_BN_BETA)) : (j == 0 ? TTK_BN_GA : (j == 1 ? TTK_BN_GMEAN : (j == 2 ? TTK_BN_GB : TTK_BN_MEAN)));
      float4 v = ld4(bnA + (size_t)row * K + c);
      if (j == 0 || j == 2) v = make_float4(v.x * sa, v.y * sa, v.z * sa, v.w * sa);
      st4(cst + i, v);
    }
    __syncthreads();
  }
#if defined(TTK_R_STAMP)
  unsigned long long st_t0, st_r0, st_wait = 0, st_aux = 0, st_loop0 = 0, st_loop1 = 0, st_x, st_y;
  TTK_STAMP(st_t0);
  TTK_RSTAMP(st_r0);
#define TTK_RB() rbarrier_s(st_wait)
#else
#define TTK_RB() rbarrier()
#endif

  if (wave >= 8) {
    // ---------------- producers: A through registers (BatchNorm form, split, ds_write), B by LDS-DMA ----------------
    __builtin_amdgcn_s_setprio(TTK_R_PPRIO);
    const int pt = tid - 512;
    const int row0 = pt >> 3, kq8 = pt & 7;  // 32 rows per pass; 8 lanes x 16 B = one 128-byte row segment
    const int sub = kq8 >> 2, chunk = (kq8 >> 1) & 1, o8 = (kq8 & 1) * 8;
    constexpr int AP = RBLK;
    // The producers' pipeline works in GROUPS of G = AP / GPS row passes of a k32 step; NSLOT register slots of one group rotate:
    // group n is converted from slot n % NSLOT and group n + NSLOT is requested into the slot that has just been freed.
    // With ONE set of a whole step (GPS = 1, NSLOT = 1: round 3) a row is requested when the producer finishes its part of step s
    // and needed when step s + 1 begins - its latency budget is the producers' barrier wait, so the loop settles at (memory
    // latency + conversion + issue) per step: 3 700 cycles measured against 2 304 of matrix work, the consumers waiting 1 000 at
    // the barrier of every step (profiles/r04_rowblock_stamps.txt).  The forward keeps two whole sets (budget: two steps); the data
    // gradient, whose rows are two tensors, three half sets (72 registers at RBLK = 6; budget: a step and a half).
    constexpr int GPS = TTK_R_SETS == 1 ? 1 : (FWD ? TTK_R_GPS_F : TTK_R_GPS_D);
    constexpr int NSLOT = TTK_R_SETS == 1 ? 1 : (FWD ? TTK_R_NSLOT_F : TTK_R_NSLOT_D);
    static_assert(AP % GPS == 0 && (GPS & (GPS - 1)) == 0, "groups per step: a power of two that divides the row passes");
    constexpr int G = AP / GPS;
    constexpr int UNR = NSLOT % GPS == 0 ? NSLOT : (GPS % NSLOT == 0 ? GPS : NSLOT * GPS);  // groups per unrolled round: slot and half are compile-time
    f32x4 ra0[NSLOT][G] = {}, ra1[NSLOT][FWD ? 1 : G] = {};
    unsigned live = 0u;  // bit i: pass i's row lies inside the row block
#pragma unroll
    for (int i = 0; i < AP; ++i) live |= (m0 + row0 + 32 * i < m_end ? 1u : 0u) << i;
    // element offset of pass 0 inside the tile's run of a channel block (ttk_common.h act_off); pass i is 32 rows = 32 kCB elements
    // further; rows past the block (the tensor's last tile may hold fewer than 32) read row 0 of the tile and become ZERO fragments below
    const unsigned aoff0 = (unsigned)(row0 * kCB + kq8 * 4);
    const float* cq = cst + kq8 * 4;
    unsigned char* wbase = lds + sub * kStr + o8;
    const int NG = nks * GPS;

    auto load_g = [&](int ks, auto hfc, auto slotc) {  // rows of group hf of step ks -> slot
      constexpr int slot = decltype(slotc)::value, hf = decltype(hfc)::value;
      // k32 step ks = channel block ks: the tile's RT x 128 contiguous bytes
      const __amdgpu_buffer_rsrc_t r0 = rbuf(A0 + ((size_t)ks * act_block_stride(M) + (size_t)m0 * kCB));
      const __amdgpu_buffer_rsrc_t r1 = rbuf((FWD ? (const T*)nullptr : A1) + ((size_t)ks * act_block_stride(M) + (size_t)m0 * kCB));
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int i = hf * G + u;
        if constexpr (kRDbg & 1) {
          f32x4 t0 = ra0[slot][u], t1 = ra1[slot][FWD ? 0 : u];  // (copies: clang rejects captured arrays as asm operands in this lambda)
          asm volatile("" : "+v"(t0), "+v"(t1));
          ra0[slot][u] = t0;
          ra1[slot][FWD ? 0 : u] = t1;
        } else {
          const unsigned o = ((live >> i) & 1u) ? aoff0 + (unsigned)(i * 32 * kCB) : (unsigned)(kq8 * 4);  // (dead rows - pass 0's too - read row 0 of the tile)
          ra0[slot][u] = rbuf_ld4<TO>(r0, o);
          if constexpr (!FWD) ra1[slot][u] = rbuf_ld4<T>(r1, o);
        }
      }
    };
    auto conv_g = [&](int ks, auto hfc, auto slotc) {  // BatchNorm form, split, ds_write of group hf of step ks
      constexpr int slot = decltype(slotc)::value, hf = decltype(hfc)::value;
      unsigned char* S = wbase + (ks & 1) * 2 * kStr;
      // this step's per-channel constants from LDS (staged once per tile with S_a already folded in: exact, a power of two)
      const float* cs = cq + ks * 32;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(cs), q1 = *reinterpret_cast<const f32x4*>(cs + K), c2 = *reinterpret_cast<const f32x4*>(cs + 2 * K);
      f32x4 q3 = c2;
      if constexpr (!FWD) q3 = *reinterpret_cast<const f32x4*>(cs + 3 * K);
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int i = hf * G + u;
        f32x4 v;
        if constexpr (FWD) {
          v = c0 * (ra0[slot][u] - q1) + c2;
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else {
          v = c0 * (ra0[slot][u] - q1) + c2 * (ra1[slot][u] - q3);
        }
        // the padding rows of a block (RT = 162 of 192) are never stored; as ZEROS they also cost the matrix pipe far less power than
        // copies of real rows would (the chip holds a higher clock on zero operands), and these kernels run at the power limit
        if (!((live >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (kRDbg & 64) {
          const f32x4 t0 = v;
          asm volatile("" ::"v"(t0));
        } else if constexpr (kRDbg & 32) {
          unsigned char* dst = S + rswz(row0 + 32 * i, chunk);
          const f32x4 w = ra0[slot][u];
          *reinterpret_cast<uint2*>(dst) = make_uint2(__float_as_uint(w.x), __float_as_uint(w.y));
          *reinterpret_cast<uint2*>(dst + APL) = make_uint2(__float_as_uint(w.z), __float_as_uint(w.w));
        } else
        rsplit_store(v, S + rswz(row0 + 32 * i, chunk), APL);
      }
    };
    // Stage s is consumed between barrier s and barrier s + 1 from slot s & 1; meanwhile the producers write the A planes of stage
    // s + 1 and the CONSUMERS' LDS-DMAs bring its weight planes.  Raw s_barrier: the producers' global loads stay in flight across it.
    static_for<NSLOT>([&](auto dc) {
      constexpr int d = decltype(dc)::value;
      if (d < NG) load_g(d / GPS, std::integral_constant<int, d % GPS>{}, dc);
    });
    for (int n0 = 0; n0 < NG; n0 += UNR) {
      static_for<UNR>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        using HF = std::integral_constant<int, d % GPS>;
        using SL = std::integral_constant<int, d % NSLOT>;
        const int n = n0 + d;
        if (n < NG) {
#if defined(TTK_R_STAMP)
          TTK_STAMP(st_x);
#endif
          conv_g(n / GPS, HF{}, SL{});
#if defined(TTK_R_STAMP)
          __builtin_amdgcn_sched_barrier(0);
          TTK_STAMP(st_y);
          if (n >= GPS) st_aux += st_y - st_x;
#endif
          // (UNR is a multiple of GPS and of NSLOT, so group n + NSLOT has half (d + NSLOT) % GPS and goes to slot d % NSLOT)
          if (n + NSLOT < NG) load_g((n + NSLOT) / GPS, std::integral_constant<int, (d + NSLOT) % GPS>{}, SL{});
          if constexpr (d % GPS == GPS - 1) {
            __builtin_amdgcn_sched_barrier(0);
            TTK_RB();  // stage n / GPS is in LDS; the consumers are done with the stage before it
#if defined(TTK_R_STAMP)
            if (n / GPS == 0) {
              TTK_STAMP(st_loop0);
              st_wait = 0;
            }
#endif
          }
        }
      });
    }
    TTK_RB();  // the consumers are done with the last stage: the ring is free for the epilogue
#if defined(TTK_R_STAMP)
    TTK_STAMP(st_loop1);
#endif
  } else {
    // ---------------- consumers: ds_read_b128 fragments + three piece products per block pair ----------------
    if constexpr (TTK_R_CPRIO != 0) __builtin_amdgcn_s_setprio(TTK_R_CPRIO);
    const int lane = tid & 63, wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, h = lane >> 5;
    constexpr bool HOLD_A = TM <= TN;  // hold the smaller fragment set of a k16 stage in registers, stream the other
    constexpr int TH = HOLD_A ? TM : TN, TS = HOLD_A ? TN : TM;
    constexpr int HPL = HOLD_A ? APL : BPL, SPL = HOLD_A ? BPL : APL;
    int hold_off[TH], strm_off[TS];
#pragma unroll
    for (int x = 0; x < TH; ++x) hold_off[x] = HOLD_A ? rswz(wm * (32 * TM) + x * 32 + r, h) : 2 * APL + rswz(wn * (32 * TN) + x * 32 + r, h);
#pragma unroll
    for (int x = 0; x < TS; ++x) strm_off[x] = HOLD_A ? 2 * APL + rswz(wn * (32 * TN) + x * 32 + r, h) : rswz(wm * (32 * TM) + x * 32 + r, h);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    f16x8 hold[TH][2] = {}, strm[2][2] = {};
    // LDS-DMA of the weight planes: a k32 step is 32 pieces of 1 KB (2 k16 stages x 2 piece planes x 8 pieces of 32 rows x 32 B); consumer
    // wave w moves pieces 4 w .. 4 w + 3 - k16 stage w >> 2, plane (w >> 1) & 1, rows 128 (w & 1) .. + 127 - right after the barrier that
    // freed the slot, and waits for them (its only vector-memory operations) before the barrier that publishes the stage.
    const int64_t bplane = (int64_t)K * Nout;
    const uint16_t* bsrc = Bq + ((wave >> 1) & 1) * bplane + ((int64_t)(wave >> 2) * Nout + n0 + (wave & 1) * 128) * 16 + lane * 8;
    const int bdst = (wave >> 2) * kStr + 2 * APL + ((wave >> 1) & 1) * BPL + (wave & 1) * 4096;
    // (inline asm: hipcc must not see a pending LDS-DMA in this wave - with the builtin it turned every counted lgkmcnt wait of the
    // fragment reads into lgkmcnt(0); M0 = the wave-uniform LDS destination, restored afterwards; the data is waited for by hand)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    auto dma_b = [&](int ks) {  // weight planes of k32 step ks -> ring slot ks & 1
      if constexpr (kRDbg & 2) return;
      const uint16_t* sp = bsrc + (int64_t)ks * 2 * Nout * 16;
      const unsigned d = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)((ks & 1) * 2 * kStr + bdst));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(sp + j * 512), "s"(d + j * 1024)
                     : "memory");
      }
    };
    dma_b(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TTK_RB();  // stage 0 is in LDS
#if defined(TTK_R_STAMP)
    TTK_STAMP(st_loop0);
    st_wait = 0;
#endif
    // Software pipeline of a k32 step (two k16 stages): the stream fragment of block x + 1 is requested before the MFMAs of block x,
    // and the hold fragments of the NEXT k16 stage replace the current ones as soon as the last block's MFMAs have read them - so
    // inside a step no MFMA waits for a read that was issued right before it; only the first fragments after the barrier are exposed.
    auto rd = [&](const unsigned char* S, int plane_bytes, int off, int p) {
      if constexpr (kRDbg & 128) {
        f16x8 z = hold[0][0];
        asm volatile("" : "+v"(z));
        return z;
      } else {
        return *reinterpret_cast<const f16x8*>(S + p * plane_bytes + off);
      }
    };
    for (int it = 0; it < nks; ++it) {
      if (it + 1 < nks) dma_b(it + 1);  // slot (it + 1) & 1 was released by the barrier just passed
      const unsigned char* S0 = lds + (it & 1) * 2 * kStr;
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int y = 0; y < TH; ++y) hold[y][p] = rd(S0, HPL, hold_off[y], p);
        strm[0][p] = rd(S0, SPL, strm_off[0], p);
      }
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        const unsigned char* S = S0 + sub * kStr;
#pragma unroll
        for (int x = 0; x < TS; ++x) {
          const int cur = (sub * TS + x) & 1;
          const bool last = x + 1 == TS;
          if (!last) {
#pragma unroll
            for (int p = 0; p < 2; ++p) strm[cur ^ 1][p] = rd(S, SPL, strm_off[x + 1], p);
          } else if (sub == 0) {
#pragma unroll
            for (int p = 0; p < 2; ++p) strm[cur ^ 1][p] = rd(S + kStr, SPL, strm_off[0], p);
          }
#pragma unroll
          for (int y = 0; y < TH; ++y) {
            // three piece products of one accumulator, smallest first: h_a l_b, l_a h_b, h_a h_b
            if constexpr (HOLD_A) {
              acc[y][x] = rmfma(hold[y][0], strm[cur][1], acc[y][x]);
              acc[y][x] = rmfma(hold[y][1], strm[cur][0], acc[y][x]);
              acc[y][x] = rmfma(hold[y][0], strm[cur][0], acc[y][x]);
            } else {
              acc[x][y] = rmfma(strm[cur][0], hold[y][1], acc[x][y]);
              acc[x][y] = rmfma(strm[cur][1], hold[y][0], acc[x][y]);
              acc[x][y] = rmfma(strm[cur][0], hold[y][0], acc[x][y]);
            }
            if (last && sub == 0) {  // hold fragment y of the second k16 stage, behind its last use in the first
#pragma unroll
              for (int p = 0; p < 2; ++p) hold[y][p] = rd(S + kStr, HPL, hold_off[y], p);
            }
          }
        }
      }
#if defined(TTK_R_STAMP)
      TTK_STAMP(st_x);
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of stage it + 1 have landed
#if defined(TTK_R_STAMP)
      TTK_STAMP(st_y);
      st_aux += st_y - st_x;
#endif
      TTK_RB();  // done with stage `it` (its fragments are in registers, its slot may be refilled); stage it + 1 is in LDS
    }
#if defined(TTK_R_STAMP)
    TTK_STAMP(st_loop1);
#endif
  }
#undef TTK_RB
  r_epilogue<RBLK, MODE, 12, T, TO>(lds, acc, out, E0, bnE, part, M, Nout, m0, m_end, n0, by, sa, sb);
#if defined(TTK_R_STAMP)
  {
    unsigned long long st_t1, st_r1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TTK_STAMP(st_t1);
    TTK_RSTAMP(st_r1);
    if ((tid & 63) == 0 && blockIdx.x < 2048) {
      unsigned long long* d = g_r_stamps + ((size_t)blockIdx.x * 12 + wave) * 8;
      d[0] = st_loop0 - st_t0; d[1] = st_loop1 - st_loop0; d[2] = st_t1 - st_loop1; d[3] = st_wait; d[4] = st_aux; d[5] = st_r1 - st_r0; d[6] = st_t1 - st_t0;
      d[7] = st_r0;
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// pw16m_k (round 4): the same tiles, LDS ring, LDS-DMA weight planes, arithmetic and epilogue as pw16r_k on EIGHT waves that each do
// both jobs - every wave owns its 32 x 32 blocks of the tile AND converts 1 / 8 of the A rows of the next stage, the conversion cut
// into RBLK parts that sit between its own MFMAs.
//
// Why (profiles/r04_rowblock_stamps.txt, r04_rowblock_gemm_timing_variants.txt): in pw16r_k a SIMD runs two MFMA waves and one
// producer wave.  With the matrix pipe kept busy by the two, the third wave's vector instructions are issued at about one per MFMA
// slot (the ~60 instructions of a step's BatchNorm form alone took 2 275 cycles beside the MFMAs, 370 without them; s_setprio
// changes nothing), so the conversion of a stage stretches over the whole step, the producers reach the barrier last and the MFMA
// waves wait 700-1 200 cycles per step for them: 3 700 cycles per k32 step against 2 304 of matrix work.  A second set of rows in
// flight did not help - the rows were there, the instructions were not issued.  Vector instructions of the wave that issues the
// MFMAs do overlap them (a handful per MFMA), so the conversion moves into the MFMA waves' own streams.
// ---------------------------------------------------------------------------------------------
template <int RBLK>
struct MGeo {
  using G = RGeo<RBLK>;
  static constexpr int NW = 8;
  static constexpr int kEpi = G::CH * G::LDC * 4 + NW * 2 * kRBN * 4;
  static constexpr int kSmem = G::kRing + G::kCst > kEpi ? G::kRing + G::kCst : kEpi;
};

#ifndef TTK_M_M16
#define TTK_M_M16 1
#endif
// M16 (default): the products on v_mfma_f32_16x16x32_f16 - a k32 step in ONE instruction per 16 x 16 block, same LDS image, same fragment
// reads, same cycles per flop as 32x32x16 - for two reasons: the 16-row blocks that lie wholly behind the end of the row block are
// skipped (RT = 162 of a 192-row tile: the last of the twelve, i.e. 8 % of the matrix work), and the guide measures the 16 x 16 shape at a
// higher clock under the power limit.  Measured: data gradient of 512 x 512 99.9 -> 97.2 us before the skip (profiles/r04_rowblock_gemm_variants.txt)
template <int RBLK, int MODE, typename T, typename TO, bool M16>
__global__ void __launch_bounds__(512) pw16m_k(const TO* __restrict__ A0, const T* __restrict__ A1, const float* __restrict__ bnA,
                                               const uint16_t* __restrict__ Bq, const float* __restrict__ wmax, TO* __restrict__ out,
                                               const T* __restrict__ E0, const float* __restrict__ bnE, float* __restrict__ part, int64_t M,
                                               int K, int Nout, int RT) {
  using G = RGeo<RBLK>;
  constexpr int RB = G::RB, WM = G::WM, WN = G::WN, TM = G::TM, TN = G::TN, APL = G::APL, BPL = G::BPL, kStr = G::kStr;
  constexpr bool FWD = MODE == RMODE_FWD;
  __shared__ __attribute__((aligned(16))) unsigned char lds[MGeo<RBLK>::kSmem];

  const int tid = threadIdx.x;
  // XCD-aware tile order (as pw16r_k): the column tiles of one row block run side by side on one L2
  const unsigned Gd = gridDim.x, Lid = blockIdx.x, NB = Nout / kRBN;
  const unsigned xq = Gd / 8, xr = Gd % 8, xcd = Lid % 8;
  const unsigned tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + Lid / 8;
  const unsigned bx = tile % NB, by = tile / NB;
  const int64_t m0 = (int64_t)by * RT;
  const int64_t m_end = m0 + RT < M ? m0 + RT : M;
  const int n0 = bx * kRBN;
  const int nks = K / 32;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float sa = pow2_scale(bnA[(size_t)TTK_BN_AUX * K + (FWD ? TTK_AUX_ACT_BOUND : TTK_AUX_DY_BOUND)]);
  const float sb = pow2_scale(*wmax);
  f32x16 acc[M16 ? 1 : TM][M16 ? 1 : TN];
  f32x4 acc16[M16 ? 2 * TM : 1][M16 ? 2 * TN : 1];
  // per-channel constants of the A operand, once per tile: [scale S_a | mean | beta S_a] (forward), [ga S_a | gmean | gb S_a | mean] (data gradient)
  float* cst = reinterpret_cast<float*>(lds + G::kRing);
  {
    constexpr int NQ = FWD ? 3 : 4;
    for (int i = tid * 4; i < NQ * K; i += 512 * 4) {
      const int j = i / K, c = i - j * K;
      const int row = FWD ? (j == 0 ? TTK_BN_SCALE : (j == 1 ? TTK_BN_MEAN : TTK_BN_BETA)) : (j == 0 ? TTK_BN_GA : (j == 1 ? TTK_BN_GMEAN : (j == 2 ? TTK_BN_GB : TTK_BN_MEAN)));
      float4 v = ld4(bnA + (size_t)row * K + c);
      if (j == 0 || j == 2) v = make_float4(v.x * sa, v.y * sa, v.z * sa, v.w * sa);
      st4(cst + i, v);
    }
    __syncthreads();
  }

  // ---- this thread's share of the A operand: 16 bytes (4 channels of the k32 step) of rows row0, row0 + 64, ... ----
  const int row0 = tid >> 3, kq8 = tid & 7;  // 64 rows per pass; 8 lanes x 16 B = one 128-byte row segment
  const int sub_ = kq8 >> 2, chunk = (kq8 >> 1) & 1, o8 = (kq8 & 1) * 8;
  constexpr int AP = RB / 64;                 // row passes per stage
  constexpr int NA = AP * (FWD ? 1 : 2);      // vector-memory loads per thread and stage
  f32x4 ra0[2][AP] = {}, ra1[2][FWD ? 1 : AP] = {};
  unsigned live = 0u;  // bit u: pass u's row lies inside the row block
#pragma unroll
  for (int u = 0; u < AP; ++u) live |= (m0 + row0 + 64 * u < m_end ? 1u : 0u) << u;
  // element offset of pass 0 inside the tile's run of a channel block; rows past the block (pass 0 too: RT may be 32) read row 0 of
  // the tile and become ZERO fragments (which also cost the matrix pipe less power than copies of real rows would)
  const unsigned aoff0 = (unsigned)(row0 * kCB + kq8 * 4);
  const float* cq = cst + kq8 * 4;
  // LDS destination of pass u in ring slot s: ((row >> 3) & 1 is that of row0: the passes are 64 rows apart)
  unsigned char* wbase = lds + sub_ * kStr + o8 + rswz(row0, chunk);

  auto load_a = [&](int ks, auto setc) {  // rows of stage ks -> register set
    constexpr int set = decltype(setc)::value;
    const __amdgpu_buffer_rsrc_t r0 = rbuf(A0 + ((size_t)ks * act_block_stride(M) + (size_t)m0 * kCB));
    const __amdgpu_buffer_rsrc_t r1 = rbuf((FWD ? (const T*)nullptr : A1) + ((size_t)ks * act_block_stride(M) + (size_t)m0 * kCB));
#pragma unroll
    for (int u = 0; u < AP; ++u) {
      if constexpr (kRDbg & 1) {
        f32x4 t0 = ra0[set][u], t1 = ra1[set][FWD ? 0 : u];
        asm volatile("" : "+v"(t0), "+v"(t1));
        ra0[set][u] = t0;
        ra1[set][FWD ? 0 : u] = t1;
      } else {
        const unsigned o = ((live >> u) & 1u) ? aoff0 + (unsigned)(u * 64 * kCB) : (unsigned)(kq8 * 4);
        ra0[set][u] = rbuf_ld4<TO>(r0, o);
        if constexpr (!FWD) ra1[set][u] = rbuf_ld4<T>(r1, o);
      }
    }
  };
  // Conversion of one row pass in two parts (a part sits behind every second MFMA triple): part 0 = BatchNorm form + high pieces, part 1
  // = low pieces + the two ds_write_b64.  `pv`, `ph` carry a pass from part 0 to part 1.
  f32x4 cc0 = {}, cq1 = {}, cc2 = {}, cq3 = {};  // the stage's per-channel constants (from LDS, S_a folded in)
  f32x4 pv = {};
  uint2 ph = make_uint2(0u, 0u);
  auto conv_consts = [&](int ks) {
    const float* cs = cq + ks * 32;
    cc0 = *reinterpret_cast<const f32x4*>(cs); cq1 = *reinterpret_cast<const f32x4*>(cs + K); cc2 = *reinterpret_cast<const f32x4*>(cs + 2 * K);
    if constexpr (!FWD) cq3 = *reinterpret_cast<const f32x4*>(cs + 3 * K);
  };
  int conv_ks = 0;
  auto conv_part = [&](auto setc, auto slotc, auto ppc) {
    constexpr int set = decltype(setc)::value, slot = decltype(slotc)::value, pp = decltype(ppc)::value, u = pp >> 1;
    if constexpr ((pp & 1) == 0) {
      // (the data gradient re-reads its four constant quads per row pass: held through the step they cost 16 registers and the
      // kernel spilled inside its loop)
      if constexpr (!FWD) conv_consts(conv_ks);
      f32x4 v;
      if constexpr (FWD) {
        v = cc0 * (ra0[set][u] - cq1) + cc2;
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      } else {
        v = cc0 * (ra0[set][u] - cq1) + cc2 * (ra1[set][u] - cq3);
      }
      if (m0 + 64 * u + 63 >= m_end) {  // (uniform: only the pass that holds the end of the row block masks its rows)
        if (!((live >> u) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      const f16x2 h01 = __builtin_convertvector(f32x2{v.x, v.y}, f16x2), h23 = __builtin_convertvector(f32x2{v.z, v.w}, f16x2);
      pv = v;
      ph = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
    } else {
#if TTK_M_MIX
      const unsigned l01 = rlow2(ph.x, pv.x, pv.y), l23 = rlow2(ph.y, pv.z, pv.w);
#else
      const f32x2 f01 = __builtin_convertvector(__builtin_bit_cast(f16x2, ph.x), f32x2), f23 = __builtin_convertvector(__builtin_bit_cast(f16x2, ph.y), f32x2);
      const unsigned l01 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{pv.x - f01.x, pv.y - f01.y}, f16x2));
      const unsigned l23 = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{pv.z - f23.x, pv.w - f23.y}, f16x2));
#endif
      unsigned char* dst = wbase + slot * 2 * kStr + u * 64 * 32;
      if constexpr (!(kRDbg & 64)) {
        *reinterpret_cast<uint2*>(dst) = ph;
        *reinterpret_cast<uint2*>(dst + APL) = make_uint2(l01, l23);
      } else {
        asm volatile("" ::"v"(l01), "v"(l23));
      }
    }
  };

  // ---- this wave's blocks of the tile ----
  const int lane = tid & 63, wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  constexpr bool HOLD_A = TM <= TN;  // hold the smaller fragment set of a k16 stage in registers, stream the other
  constexpr int TH = HOLD_A ? TM : TN, TS = HOLD_A ? TN : TM;
  constexpr int HPL = HOLD_A ? APL : BPL, SPL = HOLD_A ? BPL : APL;
  static_assert(2 * TS * TH == 2 * RBLK && 2 * AP == RBLK, "one conversion part behind every second MFMA triple");
  int hold_off[TH], strm_off[TS];
#pragma unroll
  for (int x = 0; x < TH; ++x) hold_off[x] = HOLD_A ? rswz(wm * (32 * TM) + x * 32 + r, h) : 2 * APL + rswz(wn * (32 * TN) + x * 32 + r, h);
#pragma unroll
  for (int x = 0; x < TS; ++x) strm_off[x] = HOLD_A ? 2 * APL + rswz(wn * (32 * TN) + x * 32 + r, h) : rswz(wm * (32 * TM) + x * 32 + r, h);
  if constexpr (!M16) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  } else {
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f16x8 hold[TH][2] = {}, strm[2][2] = {};
  // 16 x 16 x 32 fragments: lane l holds row l & 15, k = 8 (l >> 4) .. + 7 of the k32 step: k16 stage (l >> 5), 16-byte chunk (l >> 4) & 1
  int a16_off[M16 ? 2 * TM : 1], b16_off[M16 ? 2 * TN : 1];
  const int live_rows16 = (int)(m_end - m0) - wm * (32 * TM);  // rows of this wave's strip inside the row block
  if constexpr (M16) {
    const int r16 = lane & 15, g4 = lane >> 4;
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i) a16_off[i] = (g4 >> 1) * kStr + rswz(wm * (32 * TM) + i * 16 + r16, g4 & 1);
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j) b16_off[j] = (g4 >> 1) * kStr + 2 * APL + rswz(wn * (32 * TN) + j * 16 + r16, g4 & 1);
  }
  // LDS-DMA of the weight planes (as pw16r_k): wave w moves pieces 4 w .. 4 w + 3 of a k32 step - k16 stage w >> 2, plane (w >> 1) & 1,
  // rows 128 (w & 1) .. + 127 - right after the barrier that freed the slot, and waits for them before the barrier that publishes it.
  const int64_t bplane = (int64_t)K * Nout;
  const uint16_t* bsrc = Bq + ((wave >> 1) & 1) * bplane + ((int64_t)(wave >> 2) * Nout + n0 + (wave & 1) * 128) * 16 + lane * 8;
  const int bdst = (wave >> 2) * kStr + 2 * APL + ((wave >> 1) & 1) * BPL + (wave & 1) * 4096;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  auto dma_b = [&](int ks, int slot) {  // weight planes of k32 step ks -> ring slot
    if constexpr (kRDbg & 2) return;
    const uint16_t* sp = bsrc + (int64_t)ks * 2 * Nout * 16;
    const unsigned d = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(slot * 2 * kStr + bdst));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(sp + j * 512), "s"(d + j * 1024)
                   : "memory");
    }
  };
  auto rd = [&](const unsigned char* S, int plane_bytes, int off, int p) {
    if constexpr (kRDbg & 128) {
      f16x8 z = hold[0][0];
      asm volatile("" : "+v"(z));
      return z;
    } else {
      return *reinterpret_cast<const f16x8*>(S + p * plane_bytes + off);
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;

#if defined(TTK_R_STAMP)
  unsigned long long st_t0, st_r0, st_wait = 0, st_aux = 0, st_loop0 = 0, st_loop1 = 0, st_x, st_y;
  TTK_STAMP(st_t0);
  TTK_RSTAMP(st_r0);
#define TTK_RB() rbarrier_s(st_wait)
#else
#define TTK_RB() rbarrier()
#endif
  // ---- prologue: stage 0 into ring slot 0 (nothing to hide it under), the rows of stages 1 and 2 requested ----
  load_a(0, P0{});
  if (nks > 1) load_a(1, P1{});
  dma_b(0, 0);
  conv_consts(0);
  conv_ks = 0;
  static_for<RBLK>([&](auto ppc) { conv_part(P0{}, P0{}, ppc); });
  if (nks > 2) {
    load_a(2, P0{});
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");  // everything older than the rows of stage 2: the pieces of stage 0 have landed
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  TTK_RB();  // stage 0 is in LDS
#if defined(TTK_R_STAMP)
  TTK_STAMP(st_loop0);
  st_wait = 0;
#endif

  // One k32 step: MFMAs on stage `it` (ring slot PAR = it & 1) with, when CONV, the conversion of stage it + 1 (register set and ring
  // slot PAR ^ 1) between them; then the rows of stage it + 3 are requested into the set just freed.  Vector-memory order of a step:
  // [4 LDS-DMA pieces of stage it + 1] ... [NA rows of stage it + 3], so "all but the NA youngest" = the pieces have landed.
  auto step = [&](int it, auto parc, auto convc) {
    constexpr int PAR = decltype(parc)::value;
    constexpr bool CONV = decltype(convc)::value != 0;
    using SN = std::integral_constant<int, PAR ^ 1>;
    const unsigned char* S0 = lds + PAR * 2 * kStr;
    if constexpr (CONV) {
      dma_b(it + 1, PAR ^ 1);  // that slot was released by the barrier just passed
      conv_ks = it + 1;
      if constexpr (FWD) conv_consts(it + 1);
    }
    if constexpr (M16) {
      // hold the B fragments of the step (2 TN blocks x 2 pieces), stream the A fragments one 16-row block ahead
      f16x8 hb[2 * TN][2], sa16[2][2];
#pragma unroll
      for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) hb[j][p] = rd(S0, BPL, b16_off[j], p);
        sa16[0][p] = rd(S0, APL, a16_off[0], p);
      }
      static_for<2 * TM>([&](auto ic) {
        constexpr int i = decltype(ic)::value, cur = i & 1;
        if constexpr (i + 1 < 2 * TM) {
#pragma unroll
          for (int p = 0; p < 2; ++p) sa16[cur ^ 1][p] = rd(S0, APL, a16_off[i + 1], p);
        }
        const bool blk_live = i * 16 < live_rows16;  // (uniform) a 16-row block wholly behind the end of the row block holds zeros: no products
        static_for<2 * TN>([&](auto jc) {
          constexpr int j = decltype(jc)::value;
          if constexpr (kRDbg & 4) {
            asm volatile("" ::"v"(sa16[cur][0]), "v"(hb[j][0]));
          } else {
            if (blk_live) {
              acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sa16[cur][0], hb[j][1], acc16[i][j], 0, 0, 0);
              acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sa16[cur][1], hb[j][0], acc16[i][j], 0, 0, 0);
              acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(sa16[cur][0], hb[j][0], acc16[i][j], 0, 0, 0);
            }
          }
          constexpr int q = i * (2 * TN) + j;  // MFMA triple of the step: 4 RBLK of them, a conversion part behind every fourth
          if constexpr (CONV && (q & 3) == 3) {
            conv_part(SN{}, SN{}, std::integral_constant<int, (q >> 2)>{});
#if TTK_M_FENCE
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
        });
      });
    } else {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int y = 0; y < TH; ++y) hold[y][p] = rd(S0, HPL, hold_off[y], p);
      strm[0][p] = rd(S0, SPL, strm_off[0], p);
    }
    // software pipeline of the fragment reads as in pw16r_k: the stream fragment of block x + 1 is requested before the MFMAs of
    // block x, the hold fragments of the second k16 stage replace the first's behind their last use
    static_for<2>([&](auto subc) {
      constexpr int sub = decltype(subc)::value;
      const unsigned char* S = S0 + sub * kStr;
      static_for<TS>([&](auto xc) {
        constexpr int x = decltype(xc)::value;
        constexpr int cur = (sub * TS + x) & 1;
        constexpr bool last = x + 1 == TS;
        if constexpr (!last) {
#pragma unroll
          for (int p = 0; p < 2; ++p) strm[cur ^ 1][p] = rd(S, SPL, strm_off[x + 1], p);
        } else if constexpr (sub == 0) {
#pragma unroll
          for (int p = 0; p < 2; ++p) strm[cur ^ 1][p] = rd(S + kStr, SPL, strm_off[0], p);
        }
        static_for<TH>([&](auto yc) {
          constexpr int y = decltype(yc)::value;
          // three piece products of one accumulator, smallest first: h_a l_b, l_a h_b, h_a h_b
          if constexpr (HOLD_A) {
            acc[y][x] = rmfma(hold[y][0], strm[cur][1], acc[y][x]);
            acc[y][x] = rmfma(hold[y][1], strm[cur][0], acc[y][x]);
            acc[y][x] = rmfma(hold[y][0], strm[cur][0], acc[y][x]);
          } else {
            acc[x][y] = rmfma(strm[cur][0], hold[y][1], acc[x][y]);
            acc[x][y] = rmfma(strm[cur][1], hold[y][0], acc[x][y]);
            acc[x][y] = rmfma(strm[cur][0], hold[y][0], acc[x][y]);
          }
          if constexpr (last && sub == 0) {  // hold fragment y of the second k16 stage, behind its last use in the first
#pragma unroll
            for (int p = 0; p < 2; ++p) hold[y][p] = rd(S + kStr, HPL, hold_off[y], p);
          }
          constexpr int q = (sub * TS + x) * TH + y;  // MFMA triple of the step
          if constexpr (CONV && (q & 1) == 1) {
            conv_part(SN{}, SN{}, std::integral_constant<int, (q >> 1)>{});
#if TTK_M_FENCE
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
        });
      });
    });
    }
#if defined(TTK_R_STAMP)
    TTK_STAMP(st_x);
#endif
    if constexpr (CONV) {
      if (it + 3 < nks) {
        load_a(it + 3, SN{});
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA) : "memory");  // the pieces of stage it + 1 have landed
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
#if defined(TTK_R_STAMP)
    TTK_STAMP(st_y);
    st_aux += st_y - st_x;
#endif
    TTK_RB();  // done with stage `it` (its fragments are in registers, its slot may be refilled); stage it + 1 is in LDS
  };
  // (one loop body of two converting steps and a tail: with the step variants inside the loop hipcc kept two copies of the accumulators)
  int it = 0;
  for (; it + 2 < nks; it += 2) {
    step(it, P0{}, P1{});
    step(it + 1, P1{}, P1{});
  }
  if (it + 1 < nks) {
    step(it, P0{}, P1{});
    step(it + 1, P1{}, P0{});
  } else {
    step(it, P0{}, P0{});
  }
#if defined(TTK_R_STAMP)
  TTK_STAMP(st_loop1);
#endif
#undef TTK_RB
  if constexpr (M16) r_epilogue<RBLK, MODE, 8, T, TO, decltype(acc16), true>(lds, acc16, out, E0, bnE, part, M, Nout, m0, m_end, n0, by, sa, sb);
  else r_epilogue<RBLK, MODE, 8, T, TO>(lds, acc, out, E0, bnE, part, M, Nout, m0, m_end, n0, by, sa, sb);
#if defined(TTK_R_STAMP)
  {
    unsigned long long st_t1, st_r1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TTK_STAMP(st_t1);
    TTK_RSTAMP(st_r1);
    if ((tid & 63) == 0 && blockIdx.x < 2048) {
      unsigned long long* d = g_r_stamps + ((size_t)blockIdx.x * 12 + wave) * 8;
      d[0] = st_loop0 - st_t0; d[1] = st_loop1 - st_loop0; d[2] = st_t1 - st_loop1; d[3] = st_wait; d[4] = st_aux; d[5] = st_r1 - st_r0; d[6] = st_t1 - st_t0;
      d[7] = st_r0;
      if (wave == 0)  // (the stamp reader expects twelve waves: rows 8..11 repeat wave 0)
        for (int w = 8; w < 12; ++w)
          for (int i = 0; i < 8; ++i) g_r_stamps[((size_t)blockIdx.x * 12 + w) * 8 + i] = d[i];
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// Weight gradient  dW[co][ci] += sum_m dy[m][co] * a[m][ci]  of the layers with Cin and Cout multiples of 256, 256 x 256 tiles.
//
// The contraction runs over the pixels m, so an MFMA fragment needs 8 CONSECUTIVE m of one channel, while memory has the channels
// contiguous.  pw16_wgrad_k transposes in registers (4 rows x 4 channels per thread) and scatters 8-byte pieces into channel-major LDS
// rows - 2-way bank conflicts on every store (SQ_LDS_BANK_CONFLICT = a third of its LDS cycles) and a 128 x 256 tile per CU, i.e.
// every slice of dy is formed by two workgroups and every slice of a by four.  Here
//  * the producers store what they load: LDS holds the piece planes in the tensors' own [m][channel] order (8-byte pieces, lanes side by
//    side: conflict-free), and the CONSUMERS read their fragments transposed with ds_read_b64_tr_b16 (a 4 m x 16 channel block per
//    16-lane group, column-major into the lanes: the fragment of 32x32x16 is two such reads); rows are padded from 512 to 576 bytes so
//    that the four rows of a block fall into four different 64-byte bank windows;
//  * the tile is 256 x 256 on eight consumer waves (64 x 128 each): dy is formed twice (not per 128 output channels), a twice instead of
//    four times;
//  * every workgroup stores its tile to partial[slice][Cout][Cin] and wgrad_fold_k adds the slices in a fixed order: 64 MB of float
//    atomics per launch would take 50 us at the chip's 1.3 TB/s atomic rate, the plain stores and the fold take about half - and the
//    result is bitwise reproducible in every mode.
// ---------------------------------------------------------------------------------------------
constexpr int kTRow = 576;                       // bytes of one m-row of a piece plane: 256 channels x 2 B + 64 B (bank windows)
constexpr int kTPlane = 16 * kTRow;              // one piece plane of one operand of a k16 stage
constexpr int kTStage = 4 * kTPlane;             // [dy h][dy l][a h][a l]
constexpr int kTRing = 2 * 2 * kTStage;          // two k32 super-stages

typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8 tr_frag(const unsigned char* plane, int off) {
  // rows 8h .. 8h+3 and 8h+4 .. 8h+7 of the stage (the lane's address already holds 8h + q): element j = 4 t + q' of the fragment
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(plane + off));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(plane + off + 4 * kTRow));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(f16x8, v);
}
__device__ __forceinline__ void tsplit_store(f32x4 v, unsigned char* dst) {  // 4 consecutive channels of one m-row: h at dst, l at dst + kTPlane
  const f16x2 h01 = __builtin_convertvector(f32x2{v.x, v.y}, f16x2), h23 = __builtin_convertvector(f32x2{v.z, v.w}, f16x2);
  const f32x2 f01 = __builtin_convertvector(h01, f32x2), f23 = __builtin_convertvector(h23, f32x2);
  const f16x2 l01 = __builtin_convertvector(f32x2{v.x - f01.x, v.y - f01.y}, f16x2);
  const f16x2 l23 = __builtin_convertvector(f32x2{v.z - f23.x, v.w - f23.y}, f16x2);
  *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
  *reinterpret_cast<uint2*>(dst + kTPlane) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}

#if defined(TTK_EXPERIMENTS)  // (pw16t_wgrad_k is selectable by TTK_WGRAD_T=t only: ahead of pw16u_wgrad_k on no shape but 1024 x 1024, and there by 5 %)
// G, Y: [M][Cout] (gradient w.r.t. the BatchNorm output, raw conv output), X: [M][Cin] (raw depthwise output); partial[slice][Cout][Cin]
template <typename T, typename TG>
__global__ void __launch_bounds__(768) pw16t_wgrad_k(const TG* __restrict__ G, const T* __restrict__ Y, const float* __restrict__ bn_pw,
                                                     const T* __restrict__ X, const float* __restrict__ bn_x, float* __restrict__ partial,
                                                     int64_t M, int Cin, int Cout, int64_t rows_per_slice) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kTRing];
  const int tid = threadIdx.x;
  // XCD-aware order: every XCD gets whole slices (the tiles of a slice read the same rows of g, y and x)
  const unsigned NT = (Cout / 256) * (Cin / 256), NG = gridDim.x, Lid = blockIdx.x;
  const unsigned xq = NG / 8, xr = NG % 8, xcd = Lid % 8;
  const unsigned logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + Lid / 8;
  const unsigned tile = logical % NT, slice = logical / NT;
  const int tiles_k = Cin / 256;
  const int n0 = (tile / tiles_k) * 256, k0 = (tile % tiles_k) * 256;  // first output channel / input channel of the tile
  const int64_t m_begin = (int64_t)slice * rows_per_slice;
  const int64_t m_end = (m_begin + rows_per_slice < M) ? m_begin + rows_per_slice : M;
  const int nks = m_begin < m_end ? (int)((m_end - m_begin + 31) / 32) : 0;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float sa = pow2_scale(bn_pw[(size_t)TTK_BN_AUX * Cout + TTK_AUX_DY_BOUND]);
  const float sb = pow2_scale(bn_x[(size_t)TTK_BN_AUX * Cin + TTK_AUX_ACT_BOUND]);
  f32x16 acc[2][4];  // consumer waves: 64 output channels x 128 input channels

  if (wave >= 8) {
    // ---------------- producers: wave w owns rows 8 w .. 8 w + 7 of every k32 step, lane = channel quad ----------------
    __builtin_amdgcn_s_setprio(3);
    const int pw = wave - 8, quad = tid & 63;
    const int ca = n0 + 4 * quad, cb = k0 + 4 * quad;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_GA * Cout + ca) * sa;
    const f32x4 gb = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_GB * Cout + ca) * sa;
    const f32x4 gmean = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_GMEAN * Cout + ca);
    const f32x4 ymean = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_MEAN * Cout + ca);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(bn_x + TTK_BN_SCALE * Cin + cb) * sb;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(bn_x + TTK_BN_MEAN * Cin + cb);
    const f32x4 be = *reinterpret_cast<const f32x4*>(bn_x + TTK_BN_BETA * Cin + cb) * sb;
    f32x4 rg[8], ry[8], rx[8];
    const int64_t r0 = m_begin + 8 * pw;
    // LDS: k16 stage (pw >> 1) of the step, rows 8 (pw & 1) .. + 7
    unsigned char* wdy = lds + (pw >> 1) * kTStage + (8 * (pw & 1)) * kTRow + quad * 8;
    unsigned char* wx = wdy + 2 * kTPlane;
    auto load_row = [&](int ks, int i) {
      int64_t row = r0 + (int64_t)ks * 32 + i;
      row = row < m_end ? row : m_end - 1;  // (rows past the slice are zeroed when they are stored)
      rg[i] = rld_act4<TG>(G + act_off(row, ca, M));
      ry[i] = rld_act4<T>(Y + act_off(row, ca, M));
      rx[i] = rld_act4<T>(X + act_off(row, cb, M));
    };
    auto store = [&](int ks) {  // the eight rows of step ks: BatchNorm backward / BatchNorm + ReLU, split, to LDS
      unsigned char* dy = wdy + (ks & 1) * 2 * kTStage;
      unsigned char* xx = wx + (ks & 1) * 2 * kTStage;
      const int64_t row0 = r0 + (int64_t)ks * 32;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool live = row0 + i < m_end;  // rows past the slice contribute nothing
        f32x4 v = ga * (rg[i] - gmean) + gb * (ry[i] - ymean);
        f32x4 a = sc * (rx[i] - mu) + be;
        a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
        if (!live) { v = f32x4{0.f, 0.f, 0.f, 0.f}; a = v; }
        tsplit_store(v, dy + i * kTRow);
        tsplit_store(a, xx + i * kTRow);
      }
    };
    if (nks > 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) load_row(0, i);
      for (int s = 0; s < nks; ++s) {
        store(s);
        if (s + 1 < nks) {  // in flight across the barrier; waited for by the next store
#pragma unroll
          for (int i = 0; i < 8; ++i) load_row(s + 1, i);
        }
        rbarrier();  // stage s is in LDS; the consumers are done with stage s - 1
      }
      rbarrier();
    }
  } else {
    // ---------------- consumers: transposed fragment reads + three piece products per block pair ----------------
    const int lane = tid & 63, wm = wave >> 1, wn = wave & 1;
    const int grp = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3, h = grp >> 1;
    // this lane's address inside a plane for the 32-channel block that starts at channel `c32`: row 8h + q, channels c32 + 16 (grp & 1) + 4 p
    const int lane_off = (8 * h + q) * kTRow + (16 * (grp & 1) + 4 * p4) * 2;
    int aoff[2], boff[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) aoff[i] = lane_off + (wm * 64 + i * 32) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) boff[j] = 2 * kTPlane + lane_off + (wn * 128 + j * 32) * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (nks > 0) {
      rbarrier();  // stage 0 is in LDS
      for (int it = 0; it < nks; ++it) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          const unsigned char* S = lds + ((it & 1) * 2 + sub) * kTStage;
          f16x8 a[2][2], b[2][2];
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i][pl] = tr_frag(S + pl * kTPlane, aoff[i]);
            b[0][pl] = tr_frag(S + pl * kTPlane, boff[0]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int cur = j & 1;
            if (j + 1 < 4) {
#pragma unroll
              for (int pl = 0; pl < 2; ++pl) b[cur ^ 1][pl] = tr_frag(S + pl * kTPlane, boff[j + 1]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[cur][1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[cur][0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[cur][0], acc[i][j], 0, 0, 0);
            }
          }
        }
        rbarrier();  // done with stage `it`; stage it + 1 is in LDS
      }
    }
    // ---- the tile of this slice: plain stores (lanes = 32 consecutive input channels: 128-byte segments)
    const float inv = 1.f / (sa * sb);
    const int r = lane & 31, hh = lane >> 5;
    float* dst = partial + (size_t)slice * Cout * Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = n0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          const int col = k0 + wn * 128 + j * 32 + r;
          dst[(size_t)row * Cin + col] = acc[i][j][e] * inv;
        }
  }
}

#endif

// ---------------------------------------------------------------------------------------------
// The same weight gradient with the roles turned round: 128 (Cout) x 256 (Cin) tiles, FOUR consumer waves (64 x 128 each, one per
// SIMD) and EIGHT producer waves with TWO register sets.  Both earlier forms wait for memory in their producers: the loads of a k32
// step are issued, waited for (a round trip under load: 3-4 000 cycles), converted, and only then are the next ones issued - the
// matrix pipe (1 536 cycles per step) idles two thirds of the time (SQ_WAIT_INST_ANY 59 %).  With eight producer waves a lane holds
// 8 loads (128 B) per step instead of 24, so two steps fit into registers: the loads of steps s + 1 and s + 2 are in flight while
// step s is converted.  LDS layout, transposed fragment reads and the partial-tile + fold epilogue are those of pw16t_wgrad_k.
// ---------------------------------------------------------------------------------------------
template <typename T, typename TG>
__global__ void __launch_bounds__(768) pw16u_wgrad_k(const TG* __restrict__ G, const T* __restrict__ Y, const float* __restrict__ bn_pw,
                                                     const T* __restrict__ X, const float* __restrict__ bn_x, float* __restrict__ partial,
                                                     int64_t M, int Cin, int Cout, int64_t rows_per_slice) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kTRing];
  const int tid = threadIdx.x;
  // XCD-aware order: every XCD gets whole slices (the tiles of a slice read the same rows of g, y and x)
  const unsigned NT = (Cout / 128) * (Cin / 256), NG = gridDim.x, Lid = blockIdx.x;
  const unsigned xq = NG / 8, xr = NG % 8, xcd = Lid % 8;
  const unsigned logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + Lid / 8;
  const unsigned tile = logical % NT, slice = logical / NT;
  const int tiles_k = Cin / 256;
  const int n0 = (tile / tiles_k) * 128, k0 = (tile % tiles_k) * 256;  // first output channel / input channel of the tile
  const int64_t m_begin = (int64_t)slice * rows_per_slice;
  const int64_t m_end = (m_begin + rows_per_slice < M) ? m_begin + rows_per_slice : M;
  const int nks = m_begin < m_end ? (int)((m_end - m_begin + 31) / 32) : 0;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float sa = pow2_scale(bn_pw[(size_t)TTK_BN_AUX * Cout + TTK_AUX_DY_BOUND]);
  const float sb = pow2_scale(bn_x[(size_t)TTK_BN_AUX * Cin + TTK_AUX_ACT_BOUND]);

  if (wave >= 4) {
    // ---------------- producers: wave w owns rows 4 w .. 4 w + 3 of every k32 step.  Lane l: the x quad l (all four rows) and the dy
    // quad l & 31 (rows 2 (l >> 5), + 1)
    __builtin_amdgcn_s_setprio(3);
    const int pw = wave - 4, lane = tid & 63;
    const int qa = lane & 31, ra = 2 * (lane >> 5);
    const int ca = n0 + 4 * qa, cb = k0 + 4 * lane;
    const f32x4 ga = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_GA * Cout + ca) * sa;
    const f32x4 gb = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_GB * Cout + ca) * sa;
    const f32x4 gmean = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_GMEAN * Cout + ca);
    const f32x4 ymean = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_MEAN * Cout + ca);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(bn_x + TTK_BN_SCALE * Cin + cb) * sb;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(bn_x + TTK_BN_MEAN * Cin + cb);
    const f32x4 be = *reinterpret_cast<const f32x4*>(bn_x + TTK_BN_BETA * Cin + cb) * sb;
    f32x4 rg[2][2], ry[2][2], rx[2][4];
    const int64_t r0 = m_begin + 4 * pw;
    const size_t oa = act_off(0, ca, M), ob = act_off(0, cb, M);  // channel block of this lane's quads; rows are kCB elements apart
    // LDS: k16 stage (pw >> 2) of the step, rows 4 (pw & 3) .. + 3 of that stage
    unsigned char* wbase = lds + (pw >> 2) * kTStage + (4 * (pw & 3)) * kTRow;
    unsigned char* wdy = wbase + ra * kTRow + qa * 8;
    unsigned char* wx = wbase + 2 * kTPlane + lane * 8;
    auto load = [&](int ks, auto setc) {
      constexpr int set = decltype(setc)::value;
      const int64_t rb = r0 + (int64_t)ks * 32;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int64_t row = rb + i;
        row = row < m_end ? row : m_end - 1;  // (rows past the slice are zeroed when they are stored)
        rx[set][i] = rld_act4<T>(X + ob + (size_t)row * kCB);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int64_t row = rb + ra + j;
        row = row < m_end ? row : m_end - 1;
        rg[set][j] = rld_act4<TG>(G + oa + (size_t)row * kCB);
        ry[set][j] = rld_act4<T>(Y + oa + (size_t)row * kCB);
      }
    };
    auto store = [&](int ks, auto setc) {  // BatchNorm backward / BatchNorm + ReLU, split, to LDS
      constexpr int set = decltype(setc)::value;
      unsigned char* dy = wdy + (ks & 1) * 2 * kTStage;
      unsigned char* xx = wx + (ks & 1) * 2 * kTStage;
      const int64_t rb = r0 + (int64_t)ks * 32;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x4 v = ga * (rg[set][j] - gmean) + gb * (ry[set][j] - ymean);
        if (!(rb + ra + j < m_end)) v = f32x4{0.f, 0.f, 0.f, 0.f};  // rows past the slice contribute nothing
        tsplit_store(v, dy + j * kTRow);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        f32x4 a = sc * (rx[set][i] - mu) + be;
        a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f);
        if (!(rb + i < m_end)) a = f32x4{0.f, 0.f, 0.f, 0.f};
        tsplit_store(a, xx + i * kTRow);
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    if (nks > 0) {
      load(0, S0{});
      if (nks > 1) load(1, S1{});
      for (int s = 0; s < nks; s += 2) {
        store(s, S0{});                       // (waits for the loads of step s only: those of s + 1 stay in flight)
        if (s + 2 < nks) load(s + 2, S0{});
        rbarrier();                           // stage s is in LDS; the consumers are done with stage s - 1
        if (s + 1 < nks) {
          store(s + 1, S1{});
          if (s + 3 < nks) load(s + 3, S1{});
          rbarrier();
        }
      }
      rbarrier();
    }
  } else {
    // ---------------- consumers: transposed fragment reads + three piece products per block pair (as pw16t_wgrad_k) ----------------
    f32x16 acc[2][4];  // 64 output channels x 128 input channels
    const int lane = tid & 63, wm = wave >> 1, wn = wave & 1;
    const int grp = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3, h = grp >> 1;
    const int lane_off = (8 * h + q) * kTRow + (16 * (grp & 1) + 4 * p4) * 2;
    int aoff[2], boff[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) aoff[i] = lane_off + (wm * 64 + i * 32) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) boff[j] = 2 * kTPlane + lane_off + (wn * 128 + j * 32) * 2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (nks > 0) {
      rbarrier();  // stage 0 is in LDS
      for (int it = 0; it < nks; ++it) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
          const unsigned char* S = lds + ((it & 1) * 2 + sub) * kTStage;
          f16x8 a[2][2], b[2][2];
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i][pl] = tr_frag(S + pl * kTPlane, aoff[i]);
            b[0][pl] = tr_frag(S + pl * kTPlane, boff[0]);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int cur = j & 1;
            if (j + 1 < 4) {
#pragma unroll
              for (int pl = 0; pl < 2; ++pl) b[cur ^ 1][pl] = tr_frag(S + pl * kTPlane, boff[j + 1]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[cur][1], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][1], b[cur][0], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i][0], b[cur][0], acc[i][j], 0, 0, 0);
            }
          }
        }
        rbarrier();  // done with stage `it`; stage it + 1 is in LDS
      }
    }
    // ---- the tile of this slice: plain stores (lanes = 32 consecutive input channels: 128-byte segments; an LDS-staged form with
    // 16-byte lanes and 1 KB rows per wave measured the same: 84.6 vs 83.9 us)
    const float inv = 1.f / (sa * sb);
    const int r = lane & 31, hh = lane >> 5;
    float* dst = partial + (size_t)slice * Cout * Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = n0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hh;
          const int col = k0 + wn * 128 + j * 32 + r;
          dst[(size_t)row * Cin + col] = acc[i][j][e] * inv;
        }
  }
}

// dW[i] += partial[0][i] + partial[1][i] + ... (fixed order: bitwise reproducible); 16 B per lane
__global__ void __launch_bounds__(256) wgrad_fold_k(const float* __restrict__ partial, float* __restrict__ dW, int64_t n, int slices) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 a = ld4(dW + i);
  int s = 0;
  for (; s + 8 <= slices; s += 8) {  // eight loads in flight; the sum keeps its order (bitwise reproducible)
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld4nt(partial + (size_t)(s + u) * n + i);
#pragma unroll
    for (int u = 0; u < 8; ++u) a = add4(a, v[u]);
  }
  for (; s < slices; ++s) a = add4(a, ld4nt(partial + (size_t)s * n + i));
  st4(dW + i, a);
}

// TTK_WGRAD_T: 0 = neither transposed-read kernel; t = pw16t_wgrad_k (256 x 256 tiles) where its tile divides the shape;
// u (default) = pw16u_wgrad_k (128 x 256 tiles, eight producer waves) where its tile divides the shape
static int t_wgrad_mode() {
  static const int mode = [] { const char* e = exp_env("TTK_WGRAD_T"); return !e ? 2 : (e[0] == '0' ? 0 : (e[0] == 't' ? 1 : 2)); }();
  return mode;
}
static bool t_wgrad_wide(int Cin, int Cout) { return t_wgrad_mode() == 1 && Cin % 256 == 0 && Cout % 256 == 0; }
bool f16t_wgrad_shape(int Cin, int Cout) {
  if (t_wgrad_mode() == 0 || gemm_mode() != GEMM_F16X2 || Cin < 256 || Cin % 256) return false;
  if (t_wgrad_mode() == 1) return Cout >= 256 && Cout % 256 == 0;
  // 256 x 256 (M = 147 968 at B = 512) is HBM-bound and has two tiles only - 128 slices of partial tiles: measured 126 us against
  // 121 us of pw16_wgrad_k; the others gain 9-20 % (profiles/r03_wgrad_transposed.txt)
  return Cout >= 128 && Cout % 128 == 0 && !(Cin == 256 && Cout == 256);
}
static void t_wgrad_plan(int64_t M, int Cin, int Cout, int& tiles, int64_t& slices, int64_t& rows) {
  tiles = (Cout / (t_wgrad_wide(Cin, Cout) ? 256 : 128)) * (Cin / 256);
  slices = 256 / tiles;
  if (slices < 1) slices = 1;
  const int64_t max_slices = ceil_div(M, 64);
  if (slices > max_slices) slices = max_slices;
  rows = ceil_div(ceil_div(M, slices), 32) * 32;
  slices = ceil_div(M, rows);
}
size_t f16t_wgrad_scratch_bytes(int64_t M, int Cin, int Cout) {
  if (!f16t_wgrad_shape(Cin, Cout)) return 0;
  int tiles;
  int64_t slices, rows;
  t_wgrad_plan(M, Cin, Cout, tiles, slices, rows);
  return (size_t)slices * Cin * Cout * sizeof(float);
}
template <typename T, typename TG>
bool launch_f16t_wgrad(const TG* g, const T* y, const float* bn_pw, const T* ydw, const float* bn_dw, float* dw, float* partial, int64_t M,
                       int Cin, int Cout, hipStream_t st) {
  if (!partial || !f16t_wgrad_shape(Cin, Cout)) return false;
  int tiles;
  int64_t slices, rows;
  t_wgrad_plan(M, Cin, Cout, tiles, slices, rows);
#if defined(TTK_EXPERIMENTS)
  if (t_wgrad_wide(Cin, Cout))
    hipLaunchKernelGGL((pw16t_wgrad_k<T, TG>), dim3((unsigned)(tiles * slices)), dim3(768), 0, st, g, y, bn_pw, ydw, bn_dw, partial, M, Cin, Cout, rows);
  else
#endif
    hipLaunchKernelGGL((pw16u_wgrad_k<T, TG>), dim3((unsigned)(tiles * slices)), dim3(768), 0, st, g, y, bn_pw, ydw, bn_dw, partial, M, Cin, Cout, rows);
  const int64_t n = (int64_t)Cin * Cout;
  hipLaunchKernelGGL(wgrad_fold_k, dim3((unsigned)ceil_div(n, 1024)), dim3(256), 0, st, partial, dw, n, (int)slices);
  return true;
}
template bool launch_f16t_wgrad<float, float>(const float*, const float*, const float*, const float*, const float*, float*, float*, int64_t, int, int, hipStream_t);

// ---- tiling: row blocks of RT rows such that the tiles fill whole rounds of the CUs ------------------------------------------
bool f16r_enabled() {
  static const bool on = [] { const char* e = exp_env("TTK_GEMM_R"); return !(e && e[0] == '0'); }();
  return on && gemm_mode() == GEMM_F16X2;
}
// Shapes [M][K] x [K][Nout] that run here.  The data gradient of the 256 -> 256 layer (K = Nout = 256, M = 147 968 at B = 512) is the one
// measured slower than pw16_k's 128 x 256 tiles (147 vs 134 us: it is HBM-bound, and 193-row blocks only add epilogue time there).
bool f16r_gemm_shape(int K, int Nout, int dgrad) {
  if (!f16r_enabled() || K < 128 || K > 1024 || K % 32 != 0 || Nout < 256 || Nout % kRBN != 0) return false;
  return !(dgrad && K == 256 && Nout == 256) || exp_env("TTK_R_ALL") != nullptr;
}

struct RPlan { int rblk, rt, row_blocks; };
// Cost of a tile round in cycles per CU: a k32 step costs the larger of its matrix time (384 cycles per 32-row block)
// and the time its bytes take through the CU (A rows from HBM at ~12 B/clk, 32 KB of weight planes from L2 at ~35 B/clk); the
// epilogue writes (and, for the data gradient, reads) RT x 256 floats.
static RPlan r_plan(int64_t M, int K, int Nout) {
  const char* fe = exp_env("TTK_R_RBLK");  // experiment builds: 4 | 6 | 8 forces the tile's row blocks (anything else is ignored)
  int force = fe ? atoi(fe) : 0;
  if (force != 4 && force != 6 && force != 8) force = 0;
  static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  const int ct = Nout / kRBN, steps = K / 32;
  RPlan best{0, 0, 0};
  double best_cost = 1e300;
  for (int rblk = 4; rblk <= 8; rblk += 2) {
    if (force && rblk != force) continue;
    const int RB = 32 * rblk;
    for (int r = 1; r <= 4096; ++r) {
      const int64_t rb = (int64_t)cus * r / ct;  // row blocks that fit into r rounds
      if (rb < 1) continue;
      int64_t rt = ceil_div(M, rb);
      if (rt > RB) continue;
      if (rt < 32) rt = 32;
      const int64_t row_blocks = ceil_div(M, rt);
      const double tiles = (double)row_blocks * ct, rounds = (double)ceil_div((int64_t)tiles, cus);
      const double step = fmax(384.0 * rblk, rt * 128.0 / 12.0 + 32768.0 / 35.0);
      const double cost = rounds * (steps * step + rt * 1024.0 / 10.0 + 3000.0);
      if (cost < best_cost) { best_cost = cost; best = RPlan{rblk, (int)rt, (int)row_blocks}; }
      break;  // more rounds of smaller tiles only add per-tile overhead
    }
  }
  if (best.rblk == 0) {  // (M beyond 4 096 rounds of full tiles: full-height tiles, as many rounds as it takes - never an empty grid)
    const int rblk = force ? force : 8;
    best = RPlan{rblk, 32 * rblk, (int)ceil_div(M, 32 * rblk)};
  }
  return best;
}
int f16r_partial_rows(int64_t M, int K, int Nout, int dgrad) { return f16r_gemm_shape(K, Nout, dgrad) ? r_plan(M, K, Nout).row_blocks : 0; }
int f16r_tile_rows(int64_t M, int K, int Nout, int dgrad) { return f16r_gemm_shape(K, Nout, dgrad) ? 32 * r_plan(M, K, Nout).rblk : 0; }

// w[rows][K] fp32 -> two fp16 planes [K/16][rows][16] (16-byte chunks of a row swapped where (row >> 3) & 1) of w * pow2_scale(*wmax)
__global__ void w16r_split_k(const float* __restrict__ w, uint16_t* __restrict__ q, const float* __restrict__ wmax, int rows, int K) {
  const int64_t n = (int64_t)rows * K;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = pow2_scale(*wmax);
  const int row = (int)(i / K), k = (int)(i - (int64_t)row * K);
  const int64_t o = r_plane_index(row, k, rows);
  const float x = w[i] * s;
  const _Float16 hh = (_Float16)x;
  const _Float16 ll = (_Float16)(x - (float)hh);
  q[o] = __builtin_bit_cast(uint16_t, hh);
  q[n + o] = __builtin_bit_cast(uint16_t, ll);
}
__global__ void __launch_bounds__(256) w16r_absmax_k(const float* __restrict__ w, int64_t n, unsigned* __restrict__ wmax) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > __hip_atomic_load(wmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(wmax, __float_as_uint(m));
}

// Returns true when the shape was handled here (kernels launched on `st`).  Bm != nullptr: raw weight rows [Nout][K] that are split
// into `planes` first (per-call form, unit tests); wmax: the layer's |w| maximum (a device float the per-call form computes itself).
template <int MODE, typename T, typename TO>
bool launch_f16r_gemm(const TO* A0, const T* A1, const float* bnA, const float* Bm, TO* out, const T* E0, const float* bnE, float* part,
                      int64_t M, int K, int Nout, void* planes, float* wmax, hipStream_t st) {
  if (!planes || !wmax || !f16r_gemm_shape(K, Nout, MODE == RMODE_DGRAD)) return false;
  uint16_t* Bq = reinterpret_cast<uint16_t*>(planes);
  const int64_t nw = (int64_t)Nout * K;
  if (Bm) {
    (void)hipMemsetAsync(wmax, 0, sizeof(float), st);
    hipLaunchKernelGGL(w16r_absmax_k, dim3((unsigned)(nw / 1024 < 1 ? 1 : (nw / 1024 > 256 ? 256 : nw / 1024))), dim3(256), 0, st, Bm, nw,
                       reinterpret_cast<unsigned*>(wmax));
    hipLaunchKernelGGL(w16r_split_k, dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, st, Bm, Bq, wmax, Nout, K);
  }
  const RPlan pl = r_plan(M, K, Nout);
  const unsigned tiles = (unsigned)pl.row_blocks * (Nout / kRBN);
#define TTK_M_LAUNCH(RBLK_) \
  hipLaunchKernelGGL((pw16m_k<RBLK_, MODE, T, TO, (TTK_M_M16 != 0)>), dim3(tiles), dim3(512), 0, st, A0, A1, bnA, Bq, wmax, out, E0, bnE, part, M, K, Nout, pl.rt)
#define TTK_R_LAUNCH(RBLK_) \
  hipLaunchKernelGGL((pw16r_k<RBLK_, MODE, T, TO>), dim3(tiles), dim3(768), 0, st, A0, A1, bnA, Bq, wmax, out, E0, bnE, part, M, K, Nout, pl.rt)
  // Which form runs what, from same-box runs of the whole step (profiles/r04_rowblock_gemm_variants.txt): the eight-wave form takes the
  // data gradient (two tensors to convert per element: 0.617 vs 0.644 ms per step), the twelve-wave form keeps the forward (0.478 vs
  // 0.485) and the 256-row tiles (128 accumulator registers + two sets of rows do not fit the eight-wave form's 256).
  const bool merged = TTK_R_MERGED == 2 || (TTK_R_MERGED == 1 && MODE == RMODE_DGRAD);
  if (merged && pl.rblk == 4) TTK_M_LAUNCH(4);
  else if (merged && pl.rblk == 6) TTK_M_LAUNCH(6);
  else if (pl.rblk == 4) TTK_R_LAUNCH(4);
  else if (pl.rblk == 6) TTK_R_LAUNCH(6);
  else TTK_R_LAUNCH(8);
#undef TTK_R_LAUNCH
#undef TTK_M_LAUNCH
  return true;
}

#define TTK_RINST(T_, TG_)                                                                                                                \
  template bool launch_f16r_gemm<RMODE_FWD, T_, T_>(const T_*, const T_*, const float*, const float*, T_*, const T_*, const float*, float*, \
                                                    int64_t, int, int, void*, float*, hipStream_t);                                         \
  template bool launch_f16r_gemm<RMODE_DGRAD, T_, TG_>(const TG_*, const T_*, const float*, const float*, TG_*, const T_*, const float*,    \
                                                       float*, int64_t, int, int, void*, float*, hipStream_t);
TTK_RINST(float, float)
#undef TTK_RINST

}  // namespace ttk

#if defined(TTK_R_STAMP)
extern "C" int ttk_debug_read_r_stamps(void* host_dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ttk::g_r_stamps), bytes < sizeof(ttk::g_r_stamps) ? bytes : sizeof(ttk::g_r_stamps));
}
#endif

#if TTK_M_NOPK && defined(__HIP_DEVICE_COMPILE__)
#pragma clang attribute pop
#endif
