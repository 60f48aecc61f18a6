// Pointwise 1x1 convolutions of the wide layers - forward and data gradient - as fp32 GEMMs on the fp16 matrix pipe, "full-width" form
// (round 6).  Reference: DepthWiseBlock.conv_sep + bn_sep, backbones/mobilenet_v1.py:67-68,82-84.  Arithmetic, operand bounds and numerics
// are those of pwconv_f16.hip / pwconv_r.hip: every operand is scaled by a power of two taken from its magnitude bound and cut into two fp16
// pieces (round to nearest), a product is three v_mfma_f32_16x16x32_f16 with fp32 accumulation (h_a l_b + l_a h_b + h_a h_b).
//
// What is different from the row-block kernels (pw16r_k / pw16m_k), and why:
//  * ONE tile per workgroup and (for the 512-wide layers at B = 512) one workgroup per CU: a tile is PT = 16 PXB pixels x NT = 64 NCB
//    output channels - 176 x 512 for M = 41 472 = 256 x 162 - so the A operand (fp32 activations / gradients from HBM) is read, transformed
//    and cut into fp16 pieces ONCE (the 256-wide column tiles did all of that twice, the second time from L2), no weight slab is streamed
//    twice by a CU and there is no second round of prologue + epilogue;
//  * the orientation of the bf16-compute GEMMs (bc_gemm.hip): the WEIGHTS are the MFMA "A" operand (rows = output channels), the pixels the
//    columns, so an accumulator lane holds 4 consecutive channels of one pixel = 16 contiguous bytes of the [C/32][M][32] output: the
//    epilogue stores straight from the accumulators (64 contiguous bytes per pixel and MFMA block), no LDS round trip, no barrier;
//  * four waves, one per SIMD, each with the whole 512-entry register file: wave w owns 16 NCB channels x all pixels = NCB x PXB blocks of
//    16 x 16 (<= 88 blocks = 352 accumulator registers);
//  * the weight pieces never meet the other waves: a wave's share of a k32 step (its channel blocks' fragments, stored by
//    ttk_pwconv_prepare_weights in fragment order) reaches a wave-PRIVATE LDS ring by LDS-DMA one step ahead (no VGPRs, no barrier) and is
//    read back with lane-linear ds_read_b128;
//  * only the transformed activations are shared: all four waves convert 8-channel units of the next k32 stage (two 16-byte loads ->
//    BatchNorm form -> two fp16 pieces -> two ds_write_b128) between their MFMAs, one barrier per k32 step.
// Every vector-memory instruction of the main loop is inline asm with hand-counted s_waitcnt vmcnt(N): LDS-DMA pieces and register loads
// share the in-order vmcnt queue, and the weight pieces are requested a whole step ahead precisely so that a wait for them never has to
// wait for the (HBM-latency) activation loads that were issued behind them.
#include "ttk_common.h"
#include "conv_geom.h"
#include <atomic>
#include <type_traits>

// Timing-only bits (experiment builds, wrong results): 1 no MFMAs | 2 no activation loads | 4 no LDS-DMA | 8 no epilogue | 16 no conversion | 32 no activation fragment reads
#ifndef TTK_X_DBG
#define TTK_X_DBG 0
#endif
#ifndef TTK_X_SGB
#define TTK_X_SGB 24
#endif
#ifndef TTK_X_FENCE
#define TTK_X_FENCE 1
#endif

namespace ttk {

#if defined(TTK_X_STAMP)
// Experiment builds only (tools/exp/x_stamps.py): cycle accounting per wave - s_memtime sums of the main loop's waits, barriers and conversion spans.
__device__ unsigned long long g_x_stamps[1024 * 4 * 8];
#define TTK_XS(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define TTK_XSR(t) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
#define TTK_XSPAN(sum, ...) do { unsigned long long a_, b_; TTK_XS(a_); __VA_ARGS__; TTK_XS(b_); sum += b_ - a_; } while (0)
#else
#define TTK_XSPAN(sum, ...) do { __VA_ARGS__; } while (0)
#endif

typedef float xf32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 xf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 xf16x2 __attribute__((ext_vector_type(2)));
typedef float xf32x2 __attribute__((ext_vector_type(2)));
typedef unsigned xu32x4 __attribute__((ext_vector_type(4)));

enum { XMODE_FWD = 0, XMODE_DGRAD = 1 };
constexpr int kXDbg = TTK_X_DBG;

template <int N, typename F, int I = 0>
__device__ __forceinline__ void xfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    xfor<N, F, I + 1>(static_cast<F&&>(f));
  }
}

// ---- geometry ------------------------------------------------------------------------------------------------------------------------
template <int MODE, int NCB, int PXB>
struct XGeo {
  static constexpr int PT = 16 * PXB;                     // pixels of a tile (MFMA columns)
  static constexpr int NT = 64 * NCB;                     // output channels of a tile: 4 waves x NCB blocks of 16
  static constexpr int NU = (PXB + 3) / 4;                // conversion units (1 pixel x 8 channels) per thread and k32 stage
  static constexpr int PROWS = 64 * NU;                   // pixel rows of an LDS stage (>= PT; the surplus rows hold zeros)
  static constexpr int PLANE = PROWS * 64;                // bytes of one piece plane of a stage: 32 k x fp16 per pixel
  static constexpr int SLOT = 2 * PLANE;                  // [h plane][l plane]
  static constexpr int GCB = 2;                           // channel blocks per weight group
  static constexpr int NG = NCB / GCB;                    // weight groups per k32 step and wave
  static constexpr int GBYTES = GCB * 2 * 1024;           // a group: GCB blocks x 2 planes x (16 channels x 32 k x fp16)
  static constexpr int WRING = 4 * NG * GBYTES;           // four wave-private rings of NG groups: one k32 step ahead
  static constexpr int RL = MODE == XMODE_FWD ? 2 : 4;    // 16-byte loads per conversion unit
  static constexpr int NCONST = MODE == XMODE_FWD ? 3 : 4;
  static constexpr int kLdsFixed = 2 * SLOT + WRING;      // + NCONST * K * 4 bytes of per-channel constants
  static constexpr int UPG = (NU + NG - 1) / NG;          // conversion units handled in the head of a weight group
  static_assert(NCB % GCB == 0 && NG % 2 == 0, "group 0 of every step uses register set 0");
  static constexpr int units_in_group(int g) { return NU - g * UPG < 0 ? 0 : (NU - g * UPG < UPG ? NU - g * UPG : UPG); }
};

// LDS byte offset of the 16-byte chunk (pixel px of the stage, k chunk q) inside a piece plane: 64-byte pixel rows whose chunks are XOR-swizzled
// so that the 16x16x32 fragment read (lane = 16 q + r reads chunk q of pixel 16 pb + r: ds_read_b128, 16-lane groups) is bank-conflict free
__device__ __forceinline__ int x_achunk(int px, int q) { return px * 64 + ((q ^ ((0 - (px >> 2)) & 3)) << 4); }

// raw s_barrier fenced against compiler motion of LDS accesses; no vmcnt drain (LDS-DMA pieces and register loads stay in flight across it)
__device__ __forceinline__ void xbarrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// 16-byte streaming load of the activation operand: a BUFFER load the compiler sees (wave-uniform descriptor + 32-bit byte offset per lane).  The first
// form - an inline-asm load whose destination registers were tied to a later hand-counted wait - produced wrong tiles: between the load and the
// wait hipcc is free to copy or park "the value" of those registers (it believes the asm statement has written them), i.e. to move garbage
// around while the data is still in flight.  hipcc's own vmcnt waits for these loads do not know the LDS-DMA pieces in the queue: they
// over-wait (every piece but the youngest few must have landed too), which is harmless where the pieces are old enough (the step below).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t xbuf(const float* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ xf32x4 xld16(__amdgpu_buffer_rsrc_t r, unsigned byte_off, xf32x4 keep) {
  if constexpr (kXDbg & 2) {
    asm volatile("" : "+v"(keep) : "v"(byte_off));
    return keep;
  } else {
    return __builtin_bit_cast(xf32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 2));
  }
}
// 1 KB LDS-DMA piece: lane l's 16 bytes at sbase + voff -> LDS byte address lds_dst + 16 l (M0 saved / restored around it)
__device__ __forceinline__ void xdma16(unsigned voff, const uint16_t* sbase, unsigned lds_dst) {
  if constexpr (kXDbg & 4) return;
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
template <int N>
__device__ __forceinline__ void xwait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// low piece l = fp16(x - h) of two values whose high pieces are the halves of `h` (v_fma_mixlo/hi_f16: two instructions per pair)
__device__ __forceinline__ unsigned xlow2(unsigned h, float x0, float x1) {
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(l)
      : "v"(h), "v"(x0), "v"(x1));
  return l;
}
__device__ __forceinline__ unsigned xhigh2(float x0, float x1) {
  const xf16x2 h = __builtin_convertvector(xf32x2{x0, x1}, xf16x2);
  return __builtin_bit_cast(unsigned, h);
}

__device__ __forceinline__ xf32x4 xmfma(xf16x8 a, xf16x8 b, xf32x4 c) {
  if constexpr (kXDbg & 1) {
    asm volatile("" ::"v"(a), "v"(b));
    return c;
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
}

// The same product with the accumulator block held in VGPRs (inline asm: hipcc gives every builtin MFMA an AGPR accumulator and would move
// a VGPR-resident block into AGPRs and back around each product).  A tile of more than 64 blocks per wave (256 AGPRs) keeps the surplus blocks
// this way.  Dependent MFMAs on one accumulator need no wait states; nothing else reads these registers before the epilogue, which pads.
// All three products of a block are ONE statement that ends with the wait states an MFMA result needs before anything but the next MFMA of its
// chain may touch it: hipcc pads nothing behind an asm statement, and under register pressure it does move these registers (the first form - one
// MFMA per statement, no padding - came back with two of a quad's four values wrong in exactly the VGPR-resident blocks).
__device__ __forceinline__ void xmfma_v3(xf16x8 wh, xf16x8 wl, xf16x8 ah, xf16x8 al, xf32x4& c) {
  if constexpr (kXDbg & 1) {
    asm volatile("" : "+v"(c) : "v"(wh), "v"(wl), "v"(ah), "v"(al));
  } else {
    // (leading wait states: hipcc pads nothing in front of an asm consumer either - a register it has just restored from its AGPR spill slot by a
    // vector move must not be read by the first MFMA at once)
    asm("s_nop 3\n\tv_mfma_f32_16x16x32_f16 %0, %1, %4, %0\n\tv_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\ts_nop 9"
        : "+v"(c)
        : "v"(wh), "v"(wl), "v"(ah), "v"(al));
  }
}

// ---- the kernel ------------------------------------------------------------------------------------------------------------------------
// A0 (A1): fp32 operand tensor(s) [K/32][M][32], formed on load - forward: a = relu(scale (y - mean) + beta) from A0 = ydw and bnA = bn_dw;
// data gradient: dy = ga (g - gmean) + gb (y - mean) from A0 = g, A1 = y and bnA = bn_pw.  Wq: fragment-ordered fp16 planes of w * pow2_scale(*wmax),
// rows = OUTPUT channels (N of them), k = contraction channels.  out: [N/32][M][32].  Forward: bnE = statistics pivot [N] or NULL, sums of
// (y - pivot), (y - pivot)^2; data gradient: E0 = raw depthwise output (mask operand, [N/32][M][32]), bnE = bn_dw block, sums of g_dw and
// g_dw (ydw - mean).  part[row tile][2][N].  Tile (rt, ct): pixels [rt RT, min((rt + 1) RT, M)), channels [ct NT, + NT).
template <int MODE, int NCB, int PXB>
__global__ void __launch_bounds__(256, 1) pw16x_k(const float* __restrict__ A0, const float* __restrict__ A1, const float* __restrict__ bnA,
                                                  const uint16_t* __restrict__ Wq, const float* __restrict__ wmax, float* __restrict__ out,
                                                  const float* __restrict__ E0, const float* __restrict__ bnE, float* __restrict__ part, int64_t M,
                                                  int K, int N, int RT, int nrt, int ncol) {
  using G = XGeo<MODE, NCB, PXB>;
  constexpr bool FWD = MODE == XMODE_FWD;
  constexpr int PXA = NCB * PXB <= 64 ? PXB : 64 / NCB;  // pixel blocks whose accumulators live in AGPRs (builtin MFMAs); the rest: VGPRs (asm form)
  constexpr int PT = G::PT, NT = G::NT, NU = G::NU, PLANE = G::PLANE, SLOT = G::SLOT, GCB = G::GCB, NG = G::NG, GBYTES = G::GBYTES, RL = G::RL, UPG = G::UPG;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* const aslot = lds;                   // [2][SLOT] transformed activation stages
  unsigned char* const wring = lds + 2 * SLOT;        // [4 waves][NG groups][GBYTES]
  float* const cst = reinterpret_cast<float*>(lds + G::kLdsFixed);  // [NCONST][K]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // blocks b and b + 8 share an XCD (round-robin dispatch): the column tiles of one row tile sit there together (second reader of the A rows: L2)
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int ct = jj % ncol, rt = (jj / ncol) * 8 + xcd;
  if (rt >= nrt) return;
  const int n0 = ct * NT;
  const int64_t p0 = (int64_t)rt * RT;
  const int npx = (int)((p0 + RT < M ? p0 + RT : M) - p0);  // valid pixels of this tile (<= PT)
  const int nks = K >> 5;
  const float sa = pow2_scale(bnA[(size_t)TTK_BN_AUX * K + (FWD ? TTK_AUX_ACT_BOUND : TTK_AUX_DY_BOUND)]);
  const float sb = pow2_scale(*wmax);

  // ---- per-channel constants of the A operand, once per tile, S_a folded in: forward [scale S | mean | beta S], data gradient [ga S | gmean | gb S | mean]
  {
    for (int i = tid * 4; i < G::NCONST * K; i += 256 * 4) {
      const int j = i / K, c = i - j * K;
      const int row = FWD ? (j == 0 ? TTK_BN_SCALE : (j == 1 ? TTK_BN_MEAN : TTK_BN_BETA)) : (j == 0 ? TTK_BN_GA : (j == 1 ? TTK_BN_GMEAN : (j == 2 ? TTK_BN_GB : TTK_BN_MEAN)));
      float4 v = ld4(bnA + (size_t)row * K + c);
      if (j == 0 || j == 2) v = make_float4(v.x * sa, v.y * sa, v.z * sa, v.w * sa);
      st4(cst + i, v);
    }
  }

  // ---- conversion role: unit u of a stage = (pixel (tid >> 2) + 64 u, channels 8 q .. 8 q + 7 of the k32 block), q = tid & 3
  const int cq = tid & 3, cpx = tid >> 2;
  unsigned uoff[NU];      // byte offset of unit u's 32 bytes inside the tile's run of a channel block (clamped to the tile: every load is in bounds)
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int px = cpx + 64 * u;
    uoff[u] = (unsigned)(((px < npx ? px : 0) * 32 + 8 * cq) * 4);
  }
  const unsigned cdst = (unsigned)x_achunk(cpx, cq);  // LDS offset of unit 0 inside a plane (unit u: + 64 u x 64 B; (px >> 2) & 3 is that of cpx)
  xf32x4 raw[NU][RL];
#pragma unroll
  for (int u = 0; u < NU; ++u)
#pragma unroll
    for (int i = 0; i < RL; ++i) raw[u][i] = xf32x4{0.f, 0.f, 0.f, 0.f};
  const size_t bstride = (size_t)M * 32;  // floats between channel blocks
  const float* const a0t = A0 + (size_t)p0 * 32;
  const float* const a1t = FWD ? nullptr : A1 + (size_t)p0 * 32;
  auto load_unit = [&](int ks, auto uc) {  // requests unit u of stage ks (clamped: the surplus stages re-read the last one and are never converted)
    constexpr int u = decltype(uc)::value;
    const int kc = ks < nks ? ks : nks - 1;
    const __amdgpu_buffer_rsrc_t r0 = xbuf(a0t + (size_t)kc * bstride);
    raw[u][0] = xld16(r0, uoff[u], raw[u][0]);
    raw[u][1] = xld16(r0, uoff[u] + 16u, raw[u][1]);
    if constexpr (!FWD) {
      const __amdgpu_buffer_rsrc_t r1 = xbuf(a1t + (size_t)kc * bstride);
      raw[u][2] = xld16(r1, uoff[u], raw[u][2]);
      raw[u][3] = xld16(r1, uoff[u] + 16u, raw[u][3]);
    }
  };
  // Conversion of unit u in NP parts that the step places behind the MFMAs of successive pixel blocks (hipcc puts a conversion's ~40 vector instructions
  // in ONE block between two MFMAs otherwise: the matrix pipe idles ~250 cycles per unit).  State between the parts: cc (the stage's per-channel
  // constants of this thread's 8 channels), cv (the transformed values), chh (the high pieces).
  //   part 0: constants from LDS   1, 2: BatchNorm form of channels 0-3, 4-7   3: (ReLU) + high pieces   4: low pieces, two ds_write_b128, the unit's next loads
  // Pixels behind the tile's end need no zeroing: an MFMA column depends on its own pixel only, and those columns are neither stored nor summed
  // (their loads are clamped to the tile's first pixel: finite data).
  constexpr int NP = 5;
  xf32x4 cc[2 * G::NCONST];
  float cv[8];
  xu32x4 chh;
  auto conv_part = [&](int ks, int slot, auto uc, auto pc) {
    constexpr int u = decltype(uc)::value, part = decltype(pc)::value;
    if constexpr (kXDbg & 16) {
      if constexpr (part == 4) asm volatile("" ::"v"(raw[u][0]), "v"(raw[u][1]));
      return;
    }
    if constexpr (part == 0) {
      const float* cs = cst + ks * 32 + 8 * cq;
#pragma unroll
      for (int i = 0; i < G::NCONST; ++i) {
        cc[2 * i] = *reinterpret_cast<const xf32x4*>(cs + i * K);
        cc[2 * i + 1] = *reinterpret_cast<const xf32x4*>(cs + i * K + 4);
      }
    } else if constexpr (part == 1 || part == 2) {
      constexpr int hf = part - 1;
      xf32x4 t;
      if constexpr (FWD) t = cc[hf] * (raw[u][hf] - cc[2 + hf]) + cc[4 + hf];                                         // scale (y - mean) + beta
      else t = cc[hf] * (raw[u][hf] - cc[2 + hf]) + cc[4 + hf] * (raw[u][2 + hf] - cc[6 + hf]);                        // ga (g - gmean) + gb (y - mean)
#pragma unroll
      for (int j = 0; j < 4; ++j) cv[4 * hf + j] = t[j];
    } else if constexpr (part == 3) {
      if constexpr (FWD) {
#pragma unroll
        for (int j = 0; j < 8; ++j) cv[j] = fmaxf(cv[j], 0.f);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) chh[j] = xhigh2(cv[2 * j], cv[2 * j + 1]);
    } else {
      xu32x4 l;
#pragma unroll
      for (int j = 0; j < 4; ++j) l[j] = xlow2(chh[j], cv[2 * j], cv[2 * j + 1]);
      unsigned char* d = aslot + slot * SLOT + cdst + u * (64 * 64);
      *reinterpret_cast<xu32x4*>(d) = chh;
      *reinterpret_cast<xu32x4*>(d + PLANE) = l;
    }
  };
  auto convert_unit = [&](int ks, int slot, auto uc) {  // all parts at once (prologue)
    xfor<NP>([&](auto pc) { conv_part(ks, slot, uc, pc); });
  };

  // ---- weight pieces: wave-private ring, group g of a k32 step = channel blocks n0/16 + NCB wave + GCB g .. of BOTH planes: GCB x 2 pieces of 1 KB,
  // contiguous in the image
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  const unsigned wdst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(2 * SLOT + wave * (NG * GBYTES)));
  const uint16_t* const wsrc = Wq + ((size_t)(n0 / 16 + NCB * wave) * 2) * 512;  // + ks * (N / 16) * 1024 elements per k32 step
  const size_t wstep = (size_t)(N >> 4) * 1024;
  const unsigned wlane = (unsigned)lane * 16u;
  auto dma_group = [&](int ks, auto gc) {
    constexpr int g = decltype(gc)::value;
    const int kc = ks < nks ? ks : nks - 1;
    const uint16_t* s = wsrc + (size_t)kc * wstep + (size_t)g * (GBYTES / 2);
#pragma unroll
    for (int j = 0; j < 2 * GCB; ++j) xdma16(wlane + (unsigned)(j * 1024), s, wdst + (unsigned)(g * GBYTES + j * 1024));
  };
  const unsigned char* const wrd = wring + wave * (NG * GBYTES) + lane * 16;
  xf16x8 wf[2][GCB][2];  // [register set = group parity][block][plane]
  auto read_wgroup = [&](auto gc) {
    constexpr int g = decltype(gc)::value;
#pragma unroll
    for (int c = 0; c < GCB; ++c)
#pragma unroll
      for (int p = 0; p < 2; ++p) wf[g & 1][c][p] = *reinterpret_cast<const xf16x8*>(wrd + g * GBYTES + (c * 2 + p) * 1024);
  };

  // ---- accumulators, fragment addresses
  xf32x4 acc[NCB][PXB];
#pragma unroll
  for (int c = 0; c < NCB; ++c)
#pragma unroll
    for (int p = 0; p < PXB; ++p) acc[c][p] = xf32x4{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fq = lane >> 4;
  const int aoff = x_achunk(fr, fq);  // pixel block pb: + pb * 16 * 64 bytes
  xf16x8 af[2][2] = {};               // [double buffer][plane]
  auto read_a = [&](const unsigned char* S, int pb, int buf) {
    if constexpr (kXDbg & 32) {  // timing only: no activation fragment reads
      asm volatile("" : "+v"(af[buf][0]), "+v"(af[buf][1]));
      return;
    }
    af[buf][0] = *reinterpret_cast<const xf16x8*>(S + aoff + pb * 1024);
    af[buf][1] = *reinterpret_cast<const xf16x8*>(S + PLANE + aoff + pb * 1024);
  };

  // Vector-memory operations between an operation and the wait for it (every issue position is static; the step below):
  //   a unit's loads are waited for one step later, in the head of its group: the NG groups' pieces + the other units' loads lie between
  //   (first step: the loads come from the prologue and only the pieces of the groups before this one have been requested since);
  //   group g's pieces are waited for in the middle of the group before it, one step after their request: NG - 1 groups' pieces + the loads of
  //   the units of every group except g (whose head precedes the request of g's pieces inside group g).
  auto wait_group = [&](auto gc) {
    constexpr int g = decltype(gc)::value;
    xwait_vm<(NG - 1) * 2 * GCB + (NU - G::units_in_group(g)) * RL>();
  };

  unsigned long long st_unit = 0, st_group = 0, st_bar = 0, st_conv = 0;
#if defined(TTK_X_STAMP)
  unsigned long long st_t0, st_r0, st_l0, st_l1;
  TTK_XS(st_t0);
  TTK_XSR(st_r0);
#endif
  // ---- prologue: the weight pieces of step 0 and stage 0 of the activations; stage 0 converted into slot 0, stage 1 requested
  xfor<NG>([&](auto gc) { dma_group(0, gc); });
  xfor<NU>([&](auto uc) { load_unit(0, uc); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // the constants are in LDS
  xfor<NU>([&](auto uc) { convert_unit(0, 0, uc); });
  xfor<NU>([&](auto uc) { load_unit(1, uc); });
  read_wgroup(std::integral_constant<int, 0>{});
  xbarrier();  // stage 0 is in LDS (the fragments of group 0 are in registers: lgkmcnt(0) inside)

  // One k32 step.  Group g (static) of step ks (A slot PAR):
  //   head  for each of its units u: wait for its loads (stage ks + 1), convert into slot PAR ^ 1, request its loads of stage ks + 2;
  //   mid   the pieces of (ks + 1, g) into ring slot g (its fragments of step ks are in registers);
  //   late  wait for the pieces of the NEXT group ((ks, g + 1), or (ks + 1, 0)) and read its fragments into the other register set.
  auto step = [&](int ks, auto parc, auto firstc) {
    constexpr int PAR = decltype(parc)::value;
    constexpr bool FIRST = decltype(firstc)::value != 0;
    const unsigned char* S = aslot + PAR * SLOT;
    read_a(S, 0, 0);
    xfor<NG>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      // Positions inside a group: the conversion parts of its units spread over pixel blocks [0, PD); the pieces of (ks + 1, g) are requested at
      // block PD (an asm statement pins the LDS stores before it: the conversions are done by then, and their loads are OLDER than the request);
      // the next group's pieces are waited for and read at block PW.  hipcc's own wait at a unit's first use (part 1 of the NEXT group) finds the
      // youngest pieces PXB - PD + 1 blocks old - landed, as a rule (250-400 cycles for an L2-warm piece).
      constexpr int PD = (2 * PXB) / 3, PW = PXB - 2 > PD ? PXB - 2 : PD, NQ = UPG * NP;
      xfor<PXB>([&](auto pc) {
        constexpr int pb = decltype(pc)::value;
        constexpr int buf = (g * PXB + pb) & 1;
        if constexpr (pb + 1 < PXB) read_a(S, pb + 1, buf ^ 1);
        else if constexpr (g + 1 < NG) read_a(S, 0, buf ^ 1);
#pragma unroll
        for (int c = 0; c < GCB; ++c) {
          xf32x4& a = acc[g * GCB + c][pb];
          if constexpr (pb >= PXA) {  // VGPR-resident blocks
            xmfma_v3(wf[g & 1][c][0], wf[g & 1][c][1], af[buf][0], af[buf][1], a);
          } else {
            a = xmfma(wf[g & 1][c][0], af[buf][1], a);
            a = xmfma(wf[g & 1][c][1], af[buf][0], a);
            a = xmfma(wf[g & 1][c][0], af[buf][0], a);
          }
        }
        // conversion parts that belong behind this block's MFMAs: part q of the group's NQ = UPG * NP sits at block q * PD / NQ
        xfor<NQ>([&](auto qc) {
          constexpr int q = decltype(qc)::value, u = g * UPG + q / NP, part = q % NP;
          if constexpr (u < NU && (q * PD) / NQ == pb) {
            const int kn = ks + 1 < nks ? ks + 1 : nks - 1;  // (past the end: a rewrite of the idle slot)
#if defined(TTK_X_STAMP)
            if constexpr (part == 1) TTK_XSPAN(st_unit, asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NU - 1) * RL) : "memory"));
#endif
            TTK_XSPAN(st_conv, conv_part(kn, PAR ^ 1, std::integral_constant<int, u>{}, std::integral_constant<int, part>{});
                      if constexpr (part == NP - 1) load_unit(ks + 2, std::integral_constant<int, u>{}));
          }
        });
        if constexpr (pb == PD) dma_group(ks + 1, gc);
        if constexpr (pb == PW) {
          constexpr int gn = (g + 1) % NG;
          TTK_XSPAN(st_group, wait_group(std::integral_constant<int, gn>{}));
          read_wgroup(std::integral_constant<int, gn>{});
        }
#if TTK_X_FENCE
        __builtin_amdgcn_sched_barrier(0);
#endif
      });
    });
    TTK_XSPAN(st_bar, xbarrier());  // stage ks + 1 is in LDS, stage ks's slot may be rewritten
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
#if defined(TTK_X_STAMP)
  TTK_XS(st_l0);
#endif
  step(0, C0{}, C1{});
  int ks = 1;
  for (; ks + 2 <= nks; ks += 2) {
    step(ks, C1{}, C0{});
    step(ks + 1, C0{}, C0{});
  }
  if (ks < nks) step(ks, C1{}, C0{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the surplus requests of the last steps
#if defined(TTK_X_STAMP)
  TTK_XS(st_l1);
#endif

  // ---- epilogue: straight from the accumulators.  Block (c, pb), lane (fr, fq): channels ch .. ch + 3 of pixel p0 + 16 pb + fr.
  if constexpr (kXDbg & 8) {
#pragma unroll
    for (int c = 0; c < NCB; ++c)
#pragma unroll
      for (int p = 0; p < PXB; ++p) asm volatile("" ::"v"(acc[c][p]));
    return;
  }
  if constexpr (PXA < PXB) asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");  // the last asm-form MFMA's result -> vector ALU reads (hipcc does not see that producer)
  const float inv = 1.f / (sa * sb);  // exact: a power of two
  const int chw = n0 + 16 * NCB * wave + 4 * fq;  // first channel of block c: + 16 c
  float* const prow = part ? part + (size_t)rt * 2 * N : nullptr;
  // A channel block at a time: its PXB results are scaled into registers of their own and stored back to back (a store whose source registers are
  // rewritten by the next block's arithmetic has to have left first: with one or two temporaries hipcc serialised the stores at memory latency -
  // 13 of 87 us forward, 73 of 149 us in the data gradient, profiles/r06_fullwidth_gemm_variants.txt); the data gradient's mask operand is
  // requested a whole channel block ahead (PXB loads in flight per lane).
  xf32x4 ev[2][FWD ? 1 : PXB];
  auto cbase_of = [&](int c) { const int ch = chw + 16 * c; return ((size_t)(ch >> 5) * (size_t)M + (size_t)p0) * 32 + (ch & 31); };
  auto eload = [&](auto cc) {
    constexpr int c = decltype(cc)::value;
    if constexpr (!FWD && c < NCB) {
      const size_t cb = cbase_of(c);
#pragma unroll
      for (int pb = 0; pb < PXB; ++pb) {
        const int px = 16 * pb + fr;
        ev[c & 1][pb] = __builtin_nontemporal_load(reinterpret_cast<const xf32x4*>(E0 + cb + (size_t)(px < npx ? px : 0) * 32));  // (clamped: unconditional loads)
      }
    }
  };
  eload(std::integral_constant<int, 0>{});
  xfor<NCB>([&](auto cc) {
    constexpr int c = decltype(cc)::value;
    const int ch = chw + 16 * c;
    const size_t cbase = cbase_of(c);
    eload(std::integral_constant<int, c + 1>{});
    xf32x4 s1 = xf32x4{0.f, 0.f, 0.f, 0.f}, s2 = s1;
    xf32x4 v[PXB];
    if constexpr (FWD) {
      xf32x4 piv = xf32x4{0.f, 0.f, 0.f, 0.f};
      if (bnE) piv = *reinterpret_cast<const xf32x4*>(bnE + ch);
#pragma unroll
      for (int pb = 0; pb < PXB; ++pb) v[pb] = acc[c][pb] * inv;
#pragma unroll
      for (int pb = 0; pb < PXB; ++pb) {
        const int px = 16 * pb + fr;
        if (px < npx) {
          *reinterpret_cast<xf32x4*>(out + cbase + (size_t)px * 32) = v[pb];
          const xf32x4 d = v[pb] - piv;
          s1 += d;
          s2 += d * d;
        }
      }
    } else {
      const xf32x4 esc = *reinterpret_cast<const xf32x4*>(bnE + (size_t)TTK_BN_SCALE * N + ch);
      const xf32x4 emu = *reinterpret_cast<const xf32x4*>(bnE + (size_t)TTK_BN_MEAN * N + ch);
      const xf32x4 ebe = *reinterpret_cast<const xf32x4*>(bnE + (size_t)TTK_BN_BETA * N + ch);
#pragma unroll
      for (int pb = 0; pb < PXB; ++pb) {
        const int px = 16 * pb + fr;
        const xf32x4 yc = ev[c & 1][pb] - emu;
        const xf32x4 a = esc * yc + ebe;
        xf32x4 t = acc[c][pb] * inv;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = a[j] > 0.f ? t[j] : 0.f;
        v[pb] = t;
        if (px < npx) {
          s1 += t;
          s2 += t * yc;
        }
      }
#pragma unroll
      for (int pb = 0; pb < PXB; ++pb) {
        const int px = 16 * pb + fr;
        if (px < npx) *reinterpret_cast<xf32x4*>(out + cbase + (size_t)px * 32) = v[pb];
      }
    }
    if (prow) {  // the 16 pixel lanes of a channel quad, fixed order
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s1[j] += __shfl_xor(s1[j], off);
          s2[j] += __shfl_xor(s2[j], off);
        }
      }
      if (fr == 0) {
        *reinterpret_cast<xf32x4*>(prow + ch) = s1;
        *reinterpret_cast<xf32x4*>(prow + N + ch) = s2;
      }
    }
  });
#if defined(TTK_X_STAMP)
  {
    unsigned long long st_t1, st_r1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TTK_XS(st_t1);
    TTK_XSR(st_r1);
    if (lane == 0 && blockIdx.x < 1024) {
      unsigned long long* d = g_x_stamps + ((size_t)blockIdx.x * 4 + wave) * 8;
      d[0] = st_l0 - st_t0; d[1] = st_l1 - st_l0; d[2] = st_t1 - st_l1; d[3] = st_unit; d[4] = st_group; d[5] = st_bar; d[6] = st_conv; d[7] = st_r1 - st_r0;
    }
  }
#endif
}

// ---- host side -------------------------------------------------------------------------------------------------------------------------
// Shapes that run here (independent of M: ttk_pwconv_prepare_weights must know the image a layer's kernel reads): contraction K >= 128,
// Nout a multiple of 256.
// Round 6's measurement (profiles/r06_fullwidth_gemm_variants.txt): this form is SLOWER than the row-block kernels on every shape of the step
// (512 x 512 forward 86-90 us against 78-80, data gradient 115-118 against 96-106): it stays an experiment build (TTK_GEMM_X=1 with -DTTK_EXPERIMENTS);
// the product runs pw16r_k / pw16m_k.
bool f16x_enabled() {
  static const bool on = [] { const char* e = exp_env("TTK_GEMM_X"); return e && e[0] == '1'; }();
  return on && gemm_mode() == GEMM_F16X2;
}
bool f16x_gemm_shape(int K, int Nout, int dgrad) {
  (void)dgrad;
  return f16x_enabled() && K >= 128 && K <= 1024 && K % 32 == 0 && Nout >= 256 && Nout % 256 == 0;
}

struct XPlan { int ncb, pxb, rt, nrt, ncol; };
// Tile = 16 pxb pixels x 64 ncb channels with ncb * pxb <= 88 blocks per wave.  Nout % 512 == 0: 512-channel tiles (ncb = 8, pxb <= 11), else 256
// (ncb = 4, pxb <= 22).  The pixel count per tile is chosen so that the tiles fill whole rounds of the CUs (41 472 pixels x 512 channels: 256
// tiles of 162 pixels = one round); the instantiated pxb values bound the padding of the last pixel block.
// (19 blocks x 4 channel blocks = 304 accumulator registers spill in the data gradient's form - four loads per conversion unit: it stops at 16)
static const int kPxb8[] = {3, 6, 8, 11}, kPxb4[] = {6, 11, 13, 16, 19};
static XPlan x_plan(int64_t M, int K, int Nout, int dgrad) {
  static const int cus = [] { int dev = 0, n = 256; if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  XPlan p;
  p.ncb = Nout % 512 == 0 ? 8 : 4;
  p.ncol = Nout / (64 * p.ncb);
  const int* cand = p.ncb == 8 ? kPxb8 : kPxb4;
  const int ncand = p.ncb == 8 ? 4 : (dgrad ? 4 : 5);
  const int maxpt = 16 * cand[ncand - 1];
  // fewest rounds with full-size tiles, then the row tiles that fill them
  const int64_t rounds = ceil_div(ceil_div(M, maxpt) * p.ncol, cus);
  int64_t target = rounds * cus / p.ncol;
  if (target < 1) target = 1;
  int64_t rt = ceil_div(M, target);
  if (rt > maxpt) rt = maxpt;
  if (rt < 1) rt = 1;
  p.rt = (int)rt;
  p.nrt = (int)ceil_div(M, rt);
  p.pxb = cand[ncand - 1];
  for (int i = 0; i < ncand; ++i)
    if (16 * cand[i] >= rt) { p.pxb = cand[i]; break; }
  (void)K;
  return p;
}
int f16x_partial_rows(int64_t M, int K, int Nout, int dgrad) { return f16x_gemm_shape(K, Nout, dgrad) ? x_plan(M, K, Nout, dgrad).nrt : 0; }
int f16x_tile_rows(int64_t M, int K, int Nout, int dgrad) { return f16x_gemm_shape(K, Nout, dgrad) ? 16 * x_plan(M, K, Nout, dgrad).pxb : 0; }

template <int MODE, int NCB, int PXB>
static void x_launch(const XPlan& pl, const float* A0, const float* A1, const float* bnA, const uint16_t* Wq, const float* wmax, float* out, const float* E0,
                     const float* bnE, float* part, int64_t M, int K, int N, hipStream_t st) {
  using G = XGeo<MODE, NCB, PXB>;
  const size_t sm = (size_t)G::kLdsFixed + (size_t)G::NCONST * K * sizeof(float);
  static std::atomic<unsigned long long> done{0ull};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;
  if (dev == 63 || !(done.load(std::memory_order_relaxed) & (1ull << dev))) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(pw16x_k<MODE, NCB, PXB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess)
      done.fetch_or(1ull << dev, std::memory_order_relaxed);
  }
  const unsigned grid = (unsigned)(ceil_div(pl.nrt, 8) * 8 * pl.ncol);
  hipLaunchKernelGGL((pw16x_k<MODE, NCB, PXB>), dim3(grid), dim3(256), sm, st, A0, A1, bnA, Wq, wmax, out, E0, bnE, part, M, K, N, pl.rt, pl.nrt, pl.ncol);
}

// w[rows][K] fp32 -> the fragment-ordered image of w * pow2_scale(*wmax) (per-call form; the training step uses ttk_pwconv_prepare_weights)
__global__ void __launch_bounds__(256) w16x_split_k(const float* __restrict__ w, uint16_t* __restrict__ q, const float* __restrict__ wmax, int rows, int K) {
  const int64_t n = (int64_t)rows * K;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float s = pow2_scale(*wmax);
  const int row = (int)(i / K), k = (int)(i - (int64_t)row * K);
  const float x = w[i] * s;
  const _Float16 hh = (_Float16)x;
  const _Float16 ll = (_Float16)(x - (float)hh);
  q[x_plane_index(row, k, rows, 0)] = __builtin_bit_cast(uint16_t, hh);
  q[x_plane_index(row, k, rows, 1)] = __builtin_bit_cast(uint16_t, ll);
}
__global__ void __launch_bounds__(256) w16x_absmax_k(const float* __restrict__ w, int64_t n, unsigned* __restrict__ wmax) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > __hip_atomic_load(wmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(wmax, __float_as_uint(m));
}

// Returns true when the shape was handled here.  `planes`: the fragment-ordered image (prepared block, split code 3) - or, with Bm != nullptr (raw weight
// rows [Nout][K]: the per-call form of the unit tests), scratch that is filled first; wmax: the layer's |w| maximum (the per-call form computes it).
template <int MODE>
bool launch_f16x_gemm(const float* A0, const float* A1, const float* bnA, const float* Bm, float* out, const float* E0, const float* bnE, float* part, int64_t M,
                      int K, int Nout, void* planes, float* wmax, hipStream_t st) {
  if (!planes || !wmax || !f16x_gemm_shape(K, Nout, MODE == XMODE_DGRAD)) return false;
  const XPlan pl = x_plan(M, K, Nout, MODE == XMODE_DGRAD);
  if (Bm) {
    const int64_t nw = (int64_t)Nout * K;
    (void)hipMemsetAsync(wmax, 0, sizeof(float), st);
    hipLaunchKernelGGL(w16x_absmax_k, dim3((unsigned)(nw / 1024 < 1 ? 1 : (nw / 1024 > 256 ? 256 : nw / 1024))), dim3(256), 0, st, Bm, nw, reinterpret_cast<unsigned*>(wmax));
    hipLaunchKernelGGL(w16x_split_k, dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, st, Bm, reinterpret_cast<uint16_t*>(planes), wmax, Nout, K);
  }
  const uint16_t* Wq = reinterpret_cast<const uint16_t*>(planes);
#define TTK_X(NCB_, PXB_) \
  if (pl.ncb == NCB_ && pl.pxb == PXB_) { x_launch<MODE, NCB_, PXB_>(pl, A0, A1, bnA, Wq, wmax, out, E0, bnE, part, M, K, Nout, st); return true; }
  TTK_X(8, 3) TTK_X(8, 6) TTK_X(8, 8) TTK_X(8, 11)
  TTK_X(4, 6) TTK_X(4, 11) TTK_X(4, 13) TTK_X(4, 16)
  if constexpr (MODE == XMODE_FWD) { TTK_X(4, 19) }
#undef TTK_X
  return false;
}
template bool launch_f16x_gemm<XMODE_FWD>(const float*, const float*, const float*, const float*, float*, const float*, const float*, float*, int64_t, int, int, void*,
                                          float*, hipStream_t);
template bool launch_f16x_gemm<XMODE_DGRAD>(const float*, const float*, const float*, const float*, float*, const float*, const float*, float*, int64_t, int, int, void*,
                                            float*, hipStream_t);

}  // namespace ttk

#if defined(TTK_X_STAMP)
extern "C" int ttk_debug_read_x_stamps(void* host_dst, size_t bytes) {
  const int rc = (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ttk::g_x_stamps), bytes < sizeof(ttk::g_x_stamps) ? bytes : sizeof(ttk::g_x_stamps));
  void* dev = nullptr;  // cleared for the next launch: a smaller grid must not leave the previous launch's rows behind
  if (hipGetSymbolAddress(&dev, HIP_SYMBOL(ttk::g_x_stamps)) == hipSuccess) (void)hipMemset(dev, 0, sizeof(ttk::g_x_stamps));
  return rc;
}
#endif
