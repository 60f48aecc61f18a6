// Pointwise 1x1 convolutions of the compute-bound layers as fp32 GEMMs on the fp16 matrix pipe with a 2-piece
// round-to-nearest operand split.  Reference: DepthWiseBlock.conv_sep + bn_sep, backbones/mobilenet_v1.py:67-68,82-84.
//
// Arithmetic.  Every fp32 operand value x of a tensor with a known magnitude bound is scaled by a power of two S (exact)
// so that |x S| < 2^15 and cut into two fp16 pieces,
//     h = fp16(x S)  (round to nearest, 11 significant bits),   l = fp16(x S - h)   (the next 11 bits; x S - h is exact),
// so x S = h + l up to 2^-23 |x S|.  A product a*b is accumulated in fp32 as  h_a l_b + l_a h_b + h_a h_b  (three
// v_mfma_f32_32x32x16_f16, each piece product exact in fp32; the dropped l_a l_b is below 2^-24 |a b|) and the tile is
// multiplied by 1/(S_a S_b) on its way out.  Measured against an fp64 product this is as close as a chain of fp32 fmas
// (tests/test_pwconv_gpu.py holds every shape to that criterion; tools/exp/split16.py is the numpy model) - the same
// accuracy class as the 3-piece bf16 split of pwconv_split.hip (six products) at HALF the matrix work, two thirds of
// the LDS and L2 bytes and a cheaper conversion (v_cvt_pk_f16_f32 instead of mask/subtract chains).
//
// Range.  fp16 has 5 exponent bits: pieces below 2^-14 lose bits and anything above 65504 overflows, so each operand
// tensor carries an upper bound of its magnitude (row TTK_BN_AUX of the BatchNorm block that forms it, include/ttk.h):
// S = 2^(14 - floor(log2 bound)).  Elements down to 2^-17 of the bound keep all 22 bits; smaller ones keep an ABSOLUTE
// error of 2^-40 of the bound, far below the fp32 rounding of the elements that dominate a sum.  Bounds come from the
// statistics the step has anyway (bn.hip: Cauchy-Schwarz on the batch variance forward, the producer's max|g| backward).
//
// Structure: as pwconv_split.hip (one workgroup = 8 waves = one CU; waves 4-7 produce - global loads, BatchNorm form,
// split, ds_write into a ring of 2 x 2 k16 stages; waves 0-3 consume - ds_read_b128 fragments + MFMAs on (BM/2)x(BN/2)
// wave tiles; one s_barrier per k32), with two piece planes per operand.
#include "ttk_common.h"
#include "conv_geom.h"
#include <type_traits>

namespace ttk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { SMODE_FWD = 0, SMODE_DGRAD = 1 };

constexpr int kStage16 = 64 * (128 + 256);  // 2 planes x 32 B x (BM + BN) rows of one k16 stage
constexpr int kStride16 = kStage16 + 64;    // the two k16 halves of a producer wave's ds_write_b64 use different banks
// The LDS ring holds RS super-stages (k32) of 2 k16 stages.  RS = 3 (148 KB): a super-stage is complete one barrier
// BEFORE the consumers start on it, so they fetch its first fragments under the previous stage's last MFMAs instead of
// right after the barrier, and the producers' LDS writes of a step may trail into the next one's barrier wait.
#ifndef TTK_RS
#define TTK_RS 2
#endif
// Register sets per producer operand = steps of global loads in flight (each load has D steps to land; cycle stamps and
// the timing variants of tools/exp/variants.sh showed the producers WAITING for loads that had one step).
#ifndef TTK_D
#define TTK_D 1
#endif
#ifndef TTK_DW
#define TTK_DW 1
#endif
constexpr int ring_bytes(int rs) { return rs * 2 * kStride16; }

__device__ __forceinline__ int swz16(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 3) & 1)) << 4); }

// 2-piece split of 4 consecutive-k values (already scaled); writes the two 8-byte pieces at dst and dst + plane
__device__ __forceinline__ void split_store16(f32x4 v, unsigned char* dst, int plane) {
  const f16x2 h01 = __builtin_convertvector(f32x2{v.x, v.y}, f16x2), h23 = __builtin_convertvector(f32x2{v.z, v.w}, f16x2);
  const f32x2 f01 = __builtin_convertvector(h01, f32x2), f23 = __builtin_convertvector(h23, f32x2);
  const f16x2 l01 = __builtin_convertvector(f32x2{v.x - f01.x, v.y - f01.y}, f16x2);
  const f16x2 l23 = __builtin_convertvector(f32x2{v.z - f23.x, v.w - f23.y}, f16x2);
  *reinterpret_cast<uint2*>(dst) = make_uint2(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23));
  *reinterpret_cast<uint2*>(dst + plane) = make_uint2(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23));
}

// 4 consecutive activation values as f32x4 (streamed: non-temporal)
template <typename T>
__device__ __forceinline__ f32x4 ld_act4(const T* p) {
  const float4 v = Act<T>::ldnt(p);
  return f32x4{v.x, v.y, v.z, v.w};
}

// The consumer side of one block tile: `nks` super-stages (k32) of ds_read_b128 fragments + 3-product MFMAs into
// acc[TM][TN].  Executes exactly 1 + nks barriers (matching the producers).
template <int BM, int BN, int RS, int STRIDE = kStride16>
__device__ __forceinline__ void consume_tile16(const unsigned char* lds, int nks, int wm, int wn, int r, int h,
                                               f32x16 (&acc)[BM / 64][BN / 64]) {
  constexpr int APL = BM * 32, BPL = BN * 32;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr bool HOLD_A = TM <= TN;  // hold the smaller fragment set in registers, stream the other
  constexpr int TH = HOLD_A ? TM : TN, TS = HOLD_A ? TN : TM;
  int hold_off[TH], strm_off[TS];
#pragma unroll
  for (int x = 0; x < TH; ++x)
    hold_off[x] = HOLD_A ? swz16(wm * (BM / 2) + x * 32 + r, h) : 2 * APL + swz16(wn * (BN / 2) + x * 32 + r, h);
#pragma unroll
  for (int x = 0; x < TS; ++x)
    strm_off[x] = HOLD_A ? 2 * APL + swz16(wn * (BN / 2) + x * 32 + r, h) : swz16(wm * (BM / 2) + x * 32 + r, h);
  constexpr int HPL = HOLD_A ? APL : BPL, SPL = HOLD_A ? BPL : APL;

  f16x8 hold[TH][2], hold_n[TH][2], strm[2][2];
  auto fetch_first = [&](const unsigned char* S, f16x8 (&hd)[TH][2], f16x8 (&st)[2]) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
      for (int x = 0; x < TH; ++x) hd[x][p] = *reinterpret_cast<const f16x8*>(S + p * HPL + hold_off[x]);
      st[p] = *reinterpret_cast<const f16x8*>(S + p * SPL + strm_off[0]);
    }
  };
  __syncthreads();  // super-stage 0 (RS = 3: and 1) is in LDS
  if constexpr (RS == 3) fetch_first(lds, hold, strm[0]);
  int slot = 0;  // it % RS
  for (int it = 0; it < nks; ++it) {
    const int slot_n = slot + 1 == RS ? 0 : slot + 1;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const unsigned char* S = lds + (slot * 2 + sub) * STRIDE;
      if (RS == 2 && sub == 0) fetch_first(S, hold, strm[0]);  // first stage after the barrier: nothing could be prefetched across it
#pragma unroll
      for (int x = 0; x < TS; ++x) {
        const int cur = (sub * TS + x) & 1;
        if (x + 1 < TS) {
#pragma unroll
          for (int p = 0; p < 2; ++p) strm[cur ^ 1][p] = *reinterpret_cast<const f16x8*>(S + p * SPL + strm_off[x + 1]);
        } else if (sub == 0) {
          fetch_first(S + STRIDE, hold_n, strm[cur ^ 1]);
        } else if (RS == 3) {
          if (it + 1 < nks) fetch_first(lds + slot_n * 2 * STRIDE, hold_n, strm[cur ^ 1]);  // complete since the previous barrier
        }
        // three piece products, smallest first; (pa, pb) index the A and B pieces (0 = h, 1 = l)
#define TTK_PROD16(pa, pb)                                                                                      \
  _Pragma("unroll") for (int y = 0; y < TH; ++y) {                                                              \
  if constexpr (HOLD_A)                                                                                       \
    acc[y][x] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hold[y][pa], strm[cur][pb], acc[y][x], 0, 0, 0);       \
  else                                                                                                        \
    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(strm[cur][pa], hold[y][pb], acc[x][y], 0, 0, 0);       \
  }
        TTK_PROD16(0, 1) TTK_PROD16(1, 0) TTK_PROD16(0, 0)
#undef TTK_PROD16
      }
      if (sub == 0 || RS == 3) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int y = 0; y < TH; ++y) hold[y][p] = hold_n[y][p];
      }
    }
    __syncthreads();
    slot = slot_n;
  }
}

// The producers' schedule, shared by the three kernels: stage s is written to LDS slot s % RS from register set s % D
// (its global loads were issued D stages earlier); the consumers work on stage `it` between barrier `it` and `it + 1`
// while the producers write stage it + RS - 1.  produce(s, set) must store stage s from `set` and re-load the set for
// stage s + D; prefetch(s, set) issues the loads of the first D stages.
template <int D, int RS, typename Prefetch, typename Produce>
__device__ __forceinline__ void producer_schedule(int nks, Prefetch&& prefetch, Produce&& produce) {
  static_assert(D >= 1 && D <= 3, "register sets");
  if (0 < nks) prefetch(0, std::integral_constant<int, 0>{});
  if (D > 1 && 1 < nks) prefetch(1, std::integral_constant<int, 1 % D>{});
  if (D > 2 && 2 < nks) prefetch(2, std::integral_constant<int, 2 % D>{});
  const int total = nks + RS - 1;  // one barrier after each of s = RS-2 .. nks+RS-2: 1 + nks barriers
  for (int base = 0; base < total; base += D) {
#define TTK_SCHED_STEP(d)                                                   \
  if (D > d && base + d < total) {                                          \
    const int s_ = base + d;                                                \
    if (s_ < nks) produce(s_, std::integral_constant<int, (d) % D>{});      \
    if (s_ >= RS - 2) __syncthreads();                                      \
  }
    TTK_SCHED_STEP(0)
    TTK_SCHED_STEP(1)
    TTK_SCHED_STEP(2)
#undef TTK_SCHED_STEP
  }
}

// A: fp32 rows [M][K] (formed on load: forward relu(bn(y)), data gradient ga*(g-gmean)+gb*(y-mean)), bound in
// bnA[TTK_BN_AUX][AMODE == BNRELU ? TTK_AUX_ACT_BOUND : TTK_AUX_DY_BOUND]; Bq: two fp16 planes [K/32][Nout][32] of the
// weights scaled by pow2_scale(*wmax).
// T: storage of activations (A1 = y, E0 = mask operand); TO: storage of the A0 operand and of the output - activations in the
// forward pass, activation gradients in the data gradient
// GATHER (conv.hip, the ResNet18 variant): implicit-GEMM convolution - the A rows are gathered per tap (conv_geom.h),
// K = taps * geo.Kc, the BatchNorm block of the A operand has geo.Kc channels; AMODE_PLAIN: A0 is a materialised
// activation whose bound is *a_bound; the masked epilogue raises bnE[TTK_BN_AUX][TTK_AUX_GMAX] to max |out|.
template <int BM, int BN, int AMODE, int EMODE, int D, typename T, typename TO, bool GATHER = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(BM + BN <= 256 ? 4 : 2, BM + BN <= 256 ? 4 : 2)))
pw16_k(const TO* __restrict__ A0, const T* __restrict__ A1, const float* __restrict__ bnA,
       const uint16_t* __restrict__ Bq, const float* __restrict__ wmax, TO* __restrict__ out, const T* __restrict__ E0,
       float* __restrict__ bnE, float* __restrict__ part, int64_t M, int K, int Nout, const float* __restrict__ a_bound,
       ConvGeom geo) {
  static_assert((BM == 128 && BN == 256) || (BM == 256 && BN == 128) || (BM == 256 && BN == 64) || (BM == 128 && BN == 64), "tile shapes");
  static_assert(GATHER || (AMODE != AMODE_PLAIN && AMODE != AMODE_PLANES), "plain A operands come from the convolutions");
  constexpr bool PLANES = AMODE == AMODE_PLANES;
  constexpr int RS = TTK_RS;
  constexpr int APL = BM * 32, BPL = BN * 32;  // bytes of one piece plane of a k16 stage
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int LDC = BN + 4;
  constexpr int QN = BN / 4, RG = 512 / QN, HALVES = BM / 128, RGH = RG / HALVES, EI = 128 / RGH;
  constexpr int kEpiBytes = BM * LDC * 4 + RG * 2 * BN * 4;
  // small tiles: a smaller ring, so that two workgroups share a CU (twice the producer waves, and one's prologue / epilogue
  // under the other's main loop)
  constexpr int kStr = BM + BN <= 256 ? 64 * (BM + BN) + 64 : kStride16;
  constexpr int kSmemBytes = RS * 2 * kStr > kEpiBytes ? RS * 2 * kStr : kEpiBytes;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kSmemBytes];

  const int tid = threadIdx.x;
  // XCD-aware tile order (see pwconv.hip): every XCD gets a contiguous range of tiles.
  const unsigned G = gridDim.x, Lid = blockIdx.x, NB = Nout / BN;
  const unsigned xq = G / 8, xr = G % 8, xcd = Lid % 8;
  unsigned tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + Lid / 8;
  const int Kc = GATHER ? geo.Kc : K;  // channels per tap
  int nks = K / 32;
  // parity classes of a stride-2 data gradient (conv_geom.h): this tile's class, its pixel grid and tap set
  int ph = 0, pw = 0, Hc = geo.Hg, Wc = geo.Wg, nkw = 1;
  const bool classes = GATHER && geo.par;
  if (classes) {
    tile = Lid;  // round-robin over the XCDs: every XCD gets a mix of long and short classes
    const int c = tile >= (unsigned)geo.ctile[2] ? (tile >= (unsigned)geo.ctile[3] ? 3 : 2) : (tile >= (unsigned)geo.ctile[1] ? 1 : 0);
    tile -= geo.ctile[c];
    ph = c < 2;
    pw = !(c & 1);
    Hc = (geo.Hg - ph + 1) >> 1;
    Wc = (geo.Wg - pw + 1) >> 1;
    M = (int64_t)geo.nimg * Hc * Wc;
    nkw = geo.KW == 3 && pw ? 2 : 1;
    nks = (geo.KW == 3 && ph ? 2 : 1) * nkw * (Kc / 32);
  }
  const unsigned bx = tile % NB, by = tile / NB;
  const int64_t m0 = (int64_t)by * BM;
  const int n0 = bx * BN;
  const bool producer = __builtin_amdgcn_readfirstlane(tid) >= 256;
  const float sa = pow2_scale(AMODE == AMODE_PLAIN || PLANES ? *a_bound
                                                   : bnA[(size_t)TTK_BN_AUX * Kc + (AMODE == AMODE_BNRELU ? TTK_AUX_ACT_BOUND : TTK_AUX_DY_BOUND)]);
  const float sb = pow2_scale(*wmax);

  if (producer) {
    __builtin_amdgcn_s_setprio(3);
    const int pt = tid - 256;
    const int row0 = pt >> 3, kq8 = pt & 7;  // 32 rows per pass; 8 lanes x 16 B = one 128-byte row segment
    const int sub = kq8 >> 2, chunk = (kq8 >> 1) & 1, o8 = (kq8 & 1) * 8;
    constexpr int AP = BM / 32;
    constexpr int NQ = AMODE == AMODE_BNGRAD ? 4 : 3;
    f32x4 ra0[D][PLANES ? 1 : AP], ra1[D][AMODE == AMODE_BNGRAD ? AP : 1], q[D][NQ];
    u32x4 rp[D][PLANES ? AP : 1];  // planes: 8 k-values (16 bytes) of one piece plane per row and lane
    constexpr int BI = BN / 64;  // B rows per thread and piece plane: 64 rows x 4 chunks of 16 B (8 k) per pass
    u32x4 rb[D][2][BI];
    int64_t arow[GATHER ? 1 : AP];
    int gbase[GATHER ? AP : 1], gh[GATHER ? AP : 1], gw[GATHER ? AP : 1];  // gather: image base pixel, grid coordinates
    unsigned vmask[D];  // gather: rows whose tap falls inside the source tensor, per register set
    const int kpt = Kc / 32;  // k32 steps per tap
    if constexpr (!GATHER) {
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        int64_t row = m0 + row0 + 32 * i;
        arow[i] = (row < M ? row : M - 1) * (int64_t)kCB + kq8 * 4;  // inside a channel block (ttk_common.h act_off); clamp: rows past M are computed but never stored
      }
    } else {
      const int hw = Hc * Wc;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int64_t row = m0 + row0 + 32 * i;
        const int rr = (int)(row < M ? row : M - 1);
        const int n = rr / hw, rem = rr - n * hw;
        gh[i] = rem / Wc;
        gw[i] = rem - gh[i] * Wc;
        if (classes) { gh[i] = 2 * gh[i] + ph; gw[i] = 2 * gw[i] + pw; }
        gbase[i] = row < M ? n * geo.Hs * geo.Ws : -1;
      }
    }
    const int brow = pt >> 2, bc4 = pt & 3;
    const uint16_t* bp = Bq + (int64_t)(n0 + brow) * 32 + bc4 * 8;
    const int64_t bplane = (int64_t)K * Nout;
    unsigned char* wbase_b = lds + (bc4 >> 1) * kStr + 2 * APL;
    const float* cp = AMODE == AMODE_PLAIN || PLANES ? nullptr : bnA + kq8 * 4;
    unsigned char* wbase = lds + sub * kStr + o8;
    // planes: lane kq8 moves piece plane kq8 >> 2, k-values 8 (kq8 & 3) .. + 7 of the k32 step: k16 stage (kq8 >> 1) & 1, 16-byte half kq8 & 1
    const uint16_t* plane_src = PLANES ? (kq8 >> 2 ? reinterpret_cast<const uint16_t*>(A1) : reinterpret_cast<const uint16_t*>(A0)) : nullptr;
    unsigned char* wbase_p = lds + ((kq8 >> 1) & 1) * kStr + (kq8 >> 2) * APL;

    auto load_a = [&](int ks, auto setc) {
      constexpr int set = decltype(setc)::value;
      const int tapc = GATHER ? ks / kpt : 0, kc0 = (ks - tapc * kpt) * 32;
      int tap = tapc;
      if (classes) {  // the class's tap number -> (kh, kw): kh = 1 | {0, 2} for even | odd rows of a 3x3 kernel, 0 for 1x1
        const int jh = tapc / nkw, jw = tapc - jh * nkw;
        tap = (geo.KW == 3 ? (ph ? 2 * jh : 1) * 3 + (pw ? 2 * jw : 1) : 0);
      }
      if constexpr (!GATHER) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          ra0[set][i] = ld_act4<TO>(A0 + arow[i] + (size_t)ks * act_block_stride(M));  // k32 step ks = channel block ks
          if constexpr (AMODE == AMODE_BNGRAD) ra1[set][i] = ld_act4<T>(A1 + arow[i] + (size_t)ks * act_block_stride(M));
        }
      } else {  // the taps re-read their neighbours' rows: cached loads
        const int kh = tap / geo.KW, kw = tap - kh * geo.KW;
        unsigned vm = 0u;
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          int sh, sw;
          bool ok = gbase[i] >= 0;
          if (!geo.transposed) {
            sh = gh[i] * geo.stride - geo.pad + kh;
            sw = gw[i] * geo.stride - geo.pad + kw;
          } else {
            const int th = gh[i] + geo.pad - kh, tw = gw[i] + geo.pad - kw, sm = geo.stride - 1;  // stride 1 or 2
            ok = ok && th >= 0 && tw >= 0 && ((th | tw) & sm) == 0;
            sh = th >> sm;
            sw = tw >> sm;
          }
          ok = ok && (unsigned)sh < (unsigned)geo.Hs && (unsigned)sw < (unsigned)geo.Ws;
          if constexpr (PLANES) {
            const int64_t off = ok ? ((int64_t)(gbase[i] + sh * geo.Ws + sw) * Kc + kc0 + (kq8 & 3) * 8) : (int64_t)0;
            const u32x4 t = *reinterpret_cast<const u32x4*>(plane_src + off);
            rp[set][i] = ok ? t : u32x4{0u, 0u, 0u, 0u};  // zero padding / taps that miss the stride grid
          } else {
            const int64_t off = ok ? ((int64_t)(gbase[i] + sh * geo.Ws + sw) * Kc + kc0 + kq8 * 4) : (int64_t)(kq8 * 4);
            vm |= (ok ? 1u : 0u) << i;
            ra0[set][i] = *reinterpret_cast<const f32x4*>(A0 + off);
            if constexpr (AMODE == AMODE_BNGRAD) ra1[set][i] = *reinterpret_cast<const f32x4*>(A1 + off);
          }
        }
        vmask[set] = vm;
      }
      if constexpr (AMODE == AMODE_BNRELU) {
        q[set][0] = *reinterpret_cast<const f32x4*>(cp + TTK_BN_SCALE * Kc + kc0);
        q[set][1] = *reinterpret_cast<const f32x4*>(cp + TTK_BN_MEAN * Kc + kc0);
        q[set][2] = *reinterpret_cast<const f32x4*>(cp + TTK_BN_BETA * Kc + kc0);
      } else if constexpr (AMODE == AMODE_BNGRAD) {
        q[set][0] = *reinterpret_cast<const f32x4*>(cp + TTK_BN_GA * Kc + kc0);
        q[set][1] = *reinterpret_cast<const f32x4*>(cp + TTK_BN_GMEAN * Kc + kc0);
        q[set][2] = *reinterpret_cast<const f32x4*>(cp + TTK_BN_GB * Kc + kc0);
        q[set][3] = *reinterpret_cast<const f32x4*>(cp + TTK_BN_MEAN * Kc + kc0);
      }
    };
    auto load_b = [&](int ks, auto setc) {
      constexpr int set = decltype(setc)::value;
      int kb = ks;  // k32 step of the weight operand
      if (classes) {
        const int tapc = ks / kpt, jh = tapc / nkw, jw = tapc - jh * nkw;
        const int tap = (geo.KW == 3 ? (ph ? 2 * jh : 1) * 3 + (pw ? 2 * jw : 1) : 0);
        kb = tap * kpt + (ks - tapc * kpt);
      }
      const uint16_t* b = bp + (int64_t)kb * Nout * 32;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < BI; ++i) rb[set][p][i] = *reinterpret_cast<const u32x4*>(b + p * bplane + 64 * 32 * i);
    };
    auto store_a = [&](int ks, auto setc) {
      constexpr int set = decltype(setc)::value;
      if constexpr (PLANES) {  // no arithmetic: the 16-byte chunk straight into the ring
        unsigned char* Sp = wbase_p + (ks % RS) * 2 * kStr;
#pragma unroll
        for (int i = 0; i < AP; ++i) *reinterpret_cast<u32x4*>(Sp + swz16(row0 + 32 * i, kq8 & 1)) = rp[set][i];
        return;
      }
      unsigned char* S = wbase + (ks % RS) * 2 * kStr;
      // the scale S_a rides on the per-channel constants (exact: a power of two)
      f32x4 c0, c2;
      if constexpr (AMODE != AMODE_PLAIN) { c0 = q[set][0] * sa; c2 = q[set][2] * sa; }
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        f32x4 v;
        if constexpr (AMODE == AMODE_BNRELU) {
          v = c0 * (ra0[set][i] - q[set][1]) + c2;
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else if constexpr (AMODE == AMODE_BNGRAD) {
          v = c0 * (ra0[set][i] - q[set][1]) + c2 * (ra1[set][i] - q[set][NQ - 1]);
        } else if constexpr (AMODE == AMODE_PLAIN) {
          v = ra0[set][i] * sa;
        }
        if constexpr (GATHER)
          if (!((vmask[set] >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};  // zero padding / taps that miss the stride grid
        split_store16(v, S + swz16(row0 + 32 * i, chunk), APL);
      }
    };
    auto store_b = [&](int ks, auto setc) {  // no arithmetic: 16-byte chunks (8 k of one piece) straight into the ring
      constexpr int set = decltype(setc)::value;
      unsigned char* S = wbase_b + (ks % RS) * 2 * kStr;
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<u32x4*>(S + p * BPL + swz16(brow + 64 * i, bc4 & 1)) = rb[set][p][i];
    };
    // B is consumed first in a step and reloaded at once, then A
    producer_schedule<D, RS>(
        nks,
        [&](int s, auto setc) { load_b(s, setc); load_a(s, setc); __builtin_amdgcn_sched_barrier(0); },
        [&](int s, auto setc) {
          store_b(s, setc);
          if (s + D < nks) load_b(s + D, setc);
          __builtin_amdgcn_sched_barrier(0);
          store_a(s, setc);
          if (s + D < nks) load_a(s + D, setc);
          __builtin_amdgcn_sched_barrier(0);
        });
  } else {
    const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    consume_tile16<BM, BN, RS, kStr>(lds, nks, wm, wn, r, h, acc);
    // ---- accumulators -> LDS image [BM][LDC] (the ring is dead: the loop ended with a barrier)
    float* Cs = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const int col = wn * (BN / 2) + j * 32 + r;
          Cs[row * LDC + col] = acc[i][j][e];
        }
  }
  // ---- epilogue, all 8 waves: row-wise pass over the C image - un-scale, 16-byte stores (a wave writes 1 KB row
  // segments), the ReLU mask of the data gradient, and the BatchNorm partial sums of this tile's 128-row halves.
  // pointwise (channel-block outputs): 16 half-waves = BN/32 channel blocks x row phases, a half-wave = 4 consecutive rows x the 8 quads
  // of one block (512 contiguous bytes per instruction); convolutions (channels-last rows): a row segment per wave as before
  constexpr int NBk = BN / 32;
  const int hw_ = tid >> 5;
  const int c4 = GATHER ? tid % QN : (hw_ % NBk) * 8 + (tid & 7), rg = GATHER ? tid / QN : (hw_ / NBk) * 4 + ((tid >> 3) & 3);
  const int half = rg / RGH, rr = rg % RGH;
  const int col = n0 + 4 * c4;
  // the mask operand of the data gradient is requested before the barrier: its latency hides behind the accumulator
  // writes of the consumer waves (it was 16 dependent round trips to memory inside the loop below)
  float4 e0[EMODE == EMODE_MASK ? EI : 1];
  if constexpr (EMODE == EMODE_MASK) {
#pragma unroll
    for (int i = 0; i < EI; ++i) {
      const int64_t grow = m0 + half * 128 + rr + RGH * i;
      e0[i] = grow < M ? Act<T>::ld(E0 + (GATHER ? (size_t)grow * Nout + col : act_off(grow, col, M))) : f4(0.f);
    }
  }
  __syncthreads();
  const float inv = 1.f / (sa * sb);  // exact: a power of two
  const float* Cs = reinterpret_cast<const float*>(lds);
  float* red = reinterpret_cast<float*>(lds + BM * LDC * 4);  // [RG][2][BN]
  float4 esc = f4(0.f), emean = f4(0.f), ebeta = f4(0.f);
  if constexpr (EMODE == EMODE_MASK) {
    esc = ld4(bnE + TTK_BN_SCALE * Nout + col); emean = ld4(bnE + TTK_BN_MEAN * Nout + col); ebeta = ld4(bnE + TTK_BN_BETA * Nout + col);
  } else if constexpr (EMODE == EMODE_STATS) {
    if (bnE) emean = ld4(bnE + col);  // forward: bnE is the statistics pivot [Nout] (ttk.h), the sums are those of y - pivot
  }
  float4 s1 = f4(0.f), s2 = f4(0.f);
  float vmx = 0.f;
#pragma unroll
  for (int i = 0; i < EI; ++i) {
    const int row = half * 128 + rr + RGH * i;
    const int64_t grow = m0 + row;
    if (grow >= M) break;
    float4 v = ld4(Cs + row * LDC + 4 * c4);
    v = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
    size_t o = GATHER ? (size_t)grow * Nout + col : act_off(grow, col, M);  // convolutions (ResNet18): channels-last rows; pointwise: channel blocks
    if (classes) {  // class-local row -> pixel
      const int rr = (int)grow, hw = Hc * Wc, n = rr / hw, rem = rr - n * hw, ch = rem / Wc, cw = rem - ch * Wc;
      o = ((size_t)(n * geo.Hg + 2 * ch + ph) * geo.Wg + 2 * cw + pw) * Nout + col;
    }
    if constexpr (EMODE == EMODE_PLAIN) {
      Act<TO>::st(out + o, Act<TO>::round(v));
    } else if constexpr (EMODE == EMODE_STATS) {
      v = Act<TO>::round(v);  // statistics of what is stored
      Act<TO>::st(out + o, v);
      v = sub4(v, emean);
      s1 = add4(s1, v);
      s2 = fma4(v, v, s2);
    } else {
      const float4 yc = sub4(e0[i], emean);
      v = Act<TO>::round(mask4(v, fma4(esc, yc, ebeta)));
      Act<TO>::st(out + o, v);
      s1 = add4(s1, v);
      s2 = fma4(v, yc, s2);
      if constexpr (GATHER) vmx = fmaxf(vmx, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
  }
  if constexpr (GATHER && EMODE == EMODE_MASK) {  // the bound the next layer's split GEMMs scale this gradient by
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vmx = fmaxf(vmx, __shfl_xor(vmx, off));
    unsigned* slot = reinterpret_cast<unsigned*>(bnE + (size_t)TTK_BN_AUX * Nout + TTK_AUX_GMAX);
    if ((tid & 63) == 0 && __float_as_uint(vmx) > __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(slot, __float_as_uint(vmx));
  }
  if (part) {
    st4(red + (rg * 2 + 0) * BN + 4 * c4, s1);
    st4(red + (rg * 2 + 1) * BN + 4 * c4, s2);
    __syncthreads();
    for (int i = tid; i < HALVES * 2 * BN; i += 512) {
      const int hf = i / (2 * BN), which = (i / BN) & 1, c = i % BN;
      float a = 0.f;
      for (int qq = 0; qq < RGH; ++qq) a += red[((hf * RGH + qq) * 2 + which) * BN + c];  // fixed order: reproducible
      const int64_t prow = (int64_t)by * HALVES + hf;
      if (prow * 128 < M) part[(size_t)prow * 2 * Nout + (size_t)which * Nout + n0 + c] = a;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient  dW[co][ci] += sum_m dy[m][co] * a[m][ci]: the contraction runs over the rows m, so the MFMA
// fragments need 8 CONSECUTIVE m of one channel.  Each producer thread loads a 4 (rows) x 4 (channels) block - four
// 16-byte loads, lanes of a row group side by side in the channel direction (128-byte segments) - and the
// transposition is register naming: register e of the four rows IS the 4 consecutive m of channel e.
// grid.x = dW tiles, grid.y = slices of M.  partial == nullptr: one fp32 atomicAdd per output element and slice;
// otherwise (deterministic mode) slice s stores its tile to partial[s][Cout][Cin] and wgrad_reduce_k folds the slices
// in a fixed order.
// ---------------------------------------------------------------------------------------------
// CONV (conv.hip, the ResNet18 variant): the columns are (tap, input channel) - Cin holds their number taps * geo.Kc - and
// X is the materialised input activation [B][Hs][Ws][Kc] with bound *x_bound; the rows m enumerate OUTPUT pixels
// (Hg x Wg per image) and column (tap, ci) reads the input pixel the tap points at (zero outside).  Partial tiles
// (Cout < BM, Cin % BN != 0) are masked; dW is torch's [Cout][Kc][taps].
// APLAIN (convolutions): G already holds dy (ttk_bn_bwd_apply materialised it once for the weight and the data gradient:
// half the A bytes through the L1 and no BatchNorm arithmetic here); Y is not read.
// APLANES: G / Y are the h / l fp16 planes [M][Cout] of dy * pow2_scale(bound) (ttk_bn_bwd_apply): the A producers only
// transpose 16-bit values (v_perm) - no loads of y, no BatchNorm arithmetic, no conversion.
template <int BM, int BN, int D, typename T, typename TG, bool CONV = false, bool APLAIN = false, bool APLANES = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(CONV && BM + BN <= 256 ? 4 : 2, CONV && BM + BN <= 256 ? 4 : 2)))
pw16_wgrad_k(const TG* __restrict__ G, const T* __restrict__ Y, const float* __restrict__ bn_pw,
             const T* __restrict__ X, const float* __restrict__ bn_x, float* __restrict__ dW, float* __restrict__ partial,
             int64_t M, int Cin, int Cout, int64_t rows_per_slice, const float* __restrict__ x_bound, ConvGeom geo) {
  static_assert((BM + BN == 384 && (BM == 128 || BM == 256)) || (BM == 128 && BN == 128) || (CONV && BM == 64 && (BN == 256 || BN == 192)),
                "128x256, 256x128, 128x128 or (convolutions) 64x256, 64x192");
  constexpr int RS = TTK_RS;
  constexpr int APL = BM * 32, BPL = BN * 32;
  constexpr int TM = BM / 64, TN = BN / 64;
  // small convolution tiles: a smaller ring, so that two workgroups share a CU (twice the producer waves - their VALU work
  // bounds these tiles; measured 439 -> 312 us on ResNet layer1.  The pointwise 128x128 tile of the MobileNet backbone
  // does NOT gain: 139 -> 189 us with the 128 registers that leaves per wave)
  constexpr int kStr = CONV && BM + BN <= 256 ? 64 * (BM + BN) + 64 : kStride16;
  __shared__ __attribute__((aligned(16))) unsigned char lds[RS * 2 * kStr];

  const int tid = threadIdx.x;
  // XCD-aware order: give each XCD whole slices (all dW tiles of a slice run side by side on ONE L2, so the slice's
  // operand rows are fetched from HBM once, not once per tile).
  const unsigned NT = gridDim.x, NG = NT * gridDim.y, Lid = blockIdx.y * NT + blockIdx.x;
  const unsigned xq = NG / 8, xr = NG % 8, xcd = Lid % 8;
  const unsigned logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + Lid / 8;
  const unsigned tile = logical % NT, slice = logical / NT;
  const int tiles_k = CONV ? (Cin + BN - 1) / BN : Cin / BN;
  const int n0 = (tile / tiles_k) * BM, k0 = (tile % tiles_k) * BN;
  const int Kc = CONV ? geo.Kc : Cin;  // row length of X
  const int64_t m_begin = (int64_t)slice * rows_per_slice;
  const int64_t m_end = (m_begin + rows_per_slice < M) ? m_begin + rows_per_slice : M;
  if (m_begin >= m_end) return;  // uniform over the block, before any barrier (the host sizes the slices so that none is empty)
  const int nks = (int)((m_end - m_begin + 31) / 32);
  const bool producer = __builtin_amdgcn_readfirstlane(tid) >= 256;
  const float sa = pow2_scale(bn_pw[(size_t)TTK_BN_AUX * Cout + TTK_AUX_DY_BOUND]);
  const float sb = pow2_scale(CONV ? *x_bound : bn_x[(size_t)TTK_BN_AUX * Cin + TTK_AUX_ACT_BOUND]);

  if (producer) {
    __builtin_amdgcn_s_setprio(3);
    const int pt = tid - 256;
    const int mb = pt & 7, cq = pt >> 3;  // 8 row blocks of 4 rows x 32 channel quads per pass
    const int sub = mb >> 2, chunk = (mb >> 1) & 1, o8 = (mb & 1) * 8;
    constexpr int AP = BM >= 128 ? BM / 128 : 1, BP = (BN + 127) / 128;
    f32x4 rg[D][APLANES ? 1 : AP][4], ry[D][APLAIN || APLANES ? 1 : AP][4], rx[D][BP][4];
    u32x4 rpa[D][APLANES ? AP : 1][4];  // planes: 4 rows x 8 channels of one piece plane
    // planes: the 32 channel-quad lanes of a pass are 2 planes x 16 channel octets (64-row tiles: 2 x 8)
    constexpr int OP = (BM >= 128 ? 128 : BM) / 8;
    const int aplane = cq / OP, aoct = cq - aplane * OP;
    const uint16_t* asrc = APLANES ? (aplane ? reinterpret_cast<const uint16_t*>(Y) : reinterpret_cast<const uint16_t*>(G)) : nullptr;
    f32x4 ga[AP], gb[AP], gmean[AP], ymean[AP], sc[BP], mu[BP], be[BP];
    int ca[AP], cb[BP];
    // convolutions: tap of the B columns, validity of this thread's A rows / B columns (partial tiles), and per register
    // set the (pass, row) pairs of the in-flight B loads that hit the source tensor
    int kh[CONV ? BP : 1], kw[CONV ? BP : 1];
    bool va[AP], vb[BP];
    unsigned bmask[D];
    const bool a_on = BM >= 128 || cq < BM / 4;  // a 64-row A tile occupies half of the producers' channel quads
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      ca[p] = APLANES ? n0 + 128 * p + 8 * aoct : n0 + 4 * (cq + 32 * p);
      va[p] = true;
      if constexpr (CONV) {
        va[p] = a_on && ca[p] < Cout;
        if (!va[p]) ca[p] = 0;
      }
      ga[p] = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_GA * Cout + ca[p]) * sa;
      gb[p] = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_GB * Cout + ca[p]) * sa;
      gmean[p] = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_GMEAN * Cout + ca[p]);
      ymean[p] = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_MEAN * Cout + ca[p]);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      cb[p] = k0 + 4 * (cq + 32 * p);
      vb[p] = true;
      if constexpr (CONV) {
        vb[p] = cb[p] < Cin && 4 * (cq + 32 * p) < BN;  // (a 192-column tile ends inside the second pass)
        const int cc = vb[p] ? cb[p] : 0, tap = cc / Kc;
        cb[p] = cc - tap * Kc;
        kh[p] = tap / geo.KW;
        kw[p] = tap - kh[p] * geo.KW;
      } else {
        sc[p] = *reinterpret_cast<const f32x4*>(bn_x + TTK_BN_SCALE * Cin + cb[p]) * sb;
        mu[p] = *reinterpret_cast<const f32x4*>(bn_x + TTK_BN_MEAN * Cin + cb[p]);
        be[p] = *reinterpret_cast<const f32x4*>(bn_x + TTK_BN_BETA * Cin + cb[p]) * sb;
      }
    }
    unsigned char* wbase = lds + sub * kStr + o8;

    // Steps whose 32 rows all lie inside the slice (all but possibly the last one) take a path without row clamps and
    // zero fills.
    const int nfull = (int)((m_end - m_begin) / 32);
    // convolutions (ResNet18): channels-last rows [M][C]; pointwise: channel blocks [C/32][M][32] (ttk_common.h act_off) - elements
    // between consecutive rows, and the offset of channel c in row 0
    const int64_t rsA = CONV ? (int64_t)Cout : kCB, rsX = CONV ? (int64_t)Kc : kCB;
    auto colo = [&](int c) -> int64_t { return CONV ? (int64_t)c : (int64_t)act_off(0, c, M); };
    const TG* gp[AP];
    const T* yp[AP];
    const T* xp[BP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      gp[p] = G + (m_begin + 4 * mb) * rsA + colo(ca[p]);
      yp[p] = Y + (m_begin + 4 * mb) * rsA + colo(ca[p]);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) xp[p] = X + (m_begin + 4 * mb) * rsX + colo(cb[p]);
    // convolutions: output pixel (pn, pho, pwo) of this thread's first row of the next step, the step increments, the
    // per-column-group offset of the tap, the rows left in the slice
    int pn = 0, pho = 0, pwo = 0, dn32 = 0, dh32 = 0, dw32 = 0, prows = 0, tapoff[BP];
    if constexpr (CONV) {
      const int64_t r0 = m_begin + 4 * mb;
      const int hw = geo.Hg * geo.Wg;
      pn = (int)(r0 / hw);
      const int rem = (int)(r0 - (int64_t)pn * hw);
      pho = rem / geo.Wg;
      pwo = rem - pho * geo.Wg;
      const int q32 = 32 / geo.Wg;
      dw32 = 32 - q32 * geo.Wg;
      dn32 = q32 / geo.Hg;
      dh32 = q32 - dn32 * geo.Hg;
      prows = (int)(m_end - r0);
#pragma unroll
      for (int p = 0; p < BP; ++p) tapoff[p] = (kh[p] * geo.Ws + kw[p]) * Kc + cb[p];
    }

    auto load_a = [&](int ks, auto setc) {
      constexpr int set = decltype(setc)::value;
      if constexpr (BM < 128)
        if (!a_on) return;
      if constexpr (APLANES) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int64_t row = m_begin + (int64_t)ks * 32 + 4 * mb + i;
          row = row < m_end ? row : m_end - 1;
#pragma unroll
          for (int p = 0; p < AP; ++p) rpa[set][p][i] = *reinterpret_cast<const u32x4*>(asrc + row * Cout + ca[p]);
        }
        return;
      }
      if (ks < nfull) {
        const int64_t base = (int64_t)ks * 32 * rsA;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int p = 0; p < AP; ++p) {
            rg[set][p][i] = ld_act4<TG>(gp[p] + base + (int64_t)i * rsA);
            if constexpr (!APLAIN) ry[set][p][i] = ld_act4<T>(yp[p] + base + (int64_t)i * rsA);
          }
        return;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int64_t row = m_begin + (int64_t)ks * 32 + 4 * mb + i;
        row = row < m_end ? row : m_end - 1;
#pragma unroll
        for (int p = 0; p < AP; ++p) {
          rg[set][p][i] = ld_act4<TG>(G + row * rsA + colo(ca[p]));
          if constexpr (!APLAIN) ry[set][p][i] = ld_act4<T>(Y + row * rsA + colo(ca[p]));
        }
      }
    };
    auto load_b = [&](int ks, auto setc) {
      constexpr int set = decltype(setc)::value;
      if constexpr (CONV) {
        // The producers' instruction count IS this kernel's speed (a first version with two divisions per step,
        // 64-bit offsets and branches for the carries ran at 10 k cycles per step, 6 x the MFMA time): the pixel of
        // the thread's first row is carried from call to call (they come in step order for every D), the other three
        // rows by branch-free carries, the offsets are 32-bit (the host checks the tensor size) and split into a
        // per-row part and a per-column-group constant.
        unsigned bm = 0u;
        int wo = pwo, ho = pho, n = pn;
        const int left = prows - ks * 32;  // rows of the slice from this thread's first row of the step on
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int hr = ho * geo.stride - geo.pad, wr = wo * geo.stride - geo.pad;
          const int base = ((n * geo.Hs + hr) * geo.Ws + wr) * Kc;
#pragma unroll
          for (int p = 0; p < BP; ++p) {
            const bool ok = i < left && vb[p] && (unsigned)(hr + kh[p]) < (unsigned)geo.Hs && (unsigned)(wr + kw[p]) < (unsigned)geo.Ws;
            const unsigned off = ok ? (unsigned)(base + tapoff[p]) : (unsigned)cb[p];
            bm |= (ok ? 1u : 0u) << (4 * p + i);
            // re-read by the neighbouring taps: cached loads
            rx[set][p][i] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(X) + (size_t)(off * 4u));
          }
          ++wo;
          const bool cw = wo == geo.Wg;
          wo = cw ? 0 : wo;
          ho += cw;
          const bool ch = ho == geo.Hg;
          ho = ch ? 0 : ho;
          n += ch;
        }
        bmask[set] = bm;
        // 32 rows on: 32 = (dn32 * Hg + dh32) * Wg + dw32
        pwo += dw32;
        const bool cw = pwo >= geo.Wg;
        pwo -= cw ? geo.Wg : 0;
        pho += dh32 + cw;
        const bool ch = pho >= geo.Hg;
        pho -= ch ? geo.Hg : 0;
        pn += dn32 + ch;
        return;
      }
      if (ks < nfull) {
        const int64_t base = (int64_t)ks * 32 * rsX;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int p = 0; p < BP; ++p) rx[set][p][i] = ld_act4<T>(xp[p] + base + (int64_t)i * rsX);
        return;
      }
      const int64_t r0 = m_begin + (int64_t)ks * 32 + 4 * mb;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int64_t row = r0 + i;
        row = row < m_end ? row : m_end - 1;
#pragma unroll
        for (int p = 0; p < BP; ++p) rx[set][p][i] = ld_act4<T>(X + row * rsX + colo(cb[p]));
      }
    };
    auto store_a = [&](int ks, auto setc) {
      constexpr int set = decltype(setc)::value;
      unsigned char* S = wbase + (ks % RS) * 2 * kStr;
      const int64_t row0 = m_begin + (int64_t)ks * 32 + 4 * mb;
      const bool masked = ks >= nfull;  // uniform
      if constexpr (BM < 128)
        if (!a_on) return;
      if constexpr (APLANES) {
#pragma unroll
        for (int p = 0; p < AP; ++p) {
          u32x4 r[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) r[i] = (row0 + i < m_end && va[p]) ? rpa[set][p][i] : u32x4{0u, 0u, 0u, 0u};
#pragma unroll
          for (int d = 0; d < 4; ++d) {  // dword d = channels 2d, 2d+1 of the four rows -> per channel its four consecutive k
            const unsigned lo01 = __builtin_amdgcn_perm(r[1][d], r[0][d], 0x05040100u), lo23 = __builtin_amdgcn_perm(r[3][d], r[2][d], 0x05040100u);
            const unsigned hi01 = __builtin_amdgcn_perm(r[1][d], r[0][d], 0x07060302u), hi23 = __builtin_amdgcn_perm(r[3][d], r[2][d], 0x07060302u);
            unsigned char* dst = S + aplane * APL;
            *reinterpret_cast<uint2*>(dst + swz16(128 * p + 8 * aoct + 2 * d, chunk)) = make_uint2(lo01, lo23);
            *reinterpret_cast<uint2*>(dst + swz16(128 * p + 8 * aoct + 2 * d + 1, chunk)) = make_uint2(hi01, hi23);
          }
        }
        return;
      }
#pragma unroll
      for (int p = 0; p < AP; ++p) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (APLAIN) v[i] = rg[set][p][i] * sa;
          else v[i] = ga[p] * (rg[set][p][i] - gmean[p]) + gb[p] * (ry[set][p][i] - ymean[p]);
        }
        if (masked || (CONV && !va[p])) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (row0 + i >= m_end || !va[p]) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};  // rows past the slice contribute nothing
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          split_store16(f32x4{v[0][e], v[1][e], v[2][e], v[3][e]}, S + swz16(4 * (cq + 32 * p) + e, chunk), APL);
      }
    };
    auto store_b = [&](int ks, auto setc) {
      constexpr int set = decltype(setc)::value;
      unsigned char* S = wbase + (ks % RS) * 2 * kStr + 2 * APL;
      const int64_t row0 = m_begin + (int64_t)ks * 32 + 4 * mb;
      const bool masked = ks >= nfull;  // uniform
#pragma unroll
      for (int p = 0; p < BP; ++p) {
        if constexpr (BN % 128 != 0)
          if (4 * (cq + 32 * p) >= BN) continue;  // rows past the tile: not even zeros (the next plane starts there)
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (CONV) {
            v[i] = ((bmask[set] >> (4 * p + i)) & 1u) ? rx[set][p][i] * sb : f32x4{0.f, 0.f, 0.f, 0.f};
          } else {
            v[i] = sc[p] * (rx[set][p][i] - mu[p]) + be[p];
            v[i].x = fmaxf(v[i].x, 0.f); v[i].y = fmaxf(v[i].y, 0.f); v[i].z = fmaxf(v[i].z, 0.f); v[i].w = fmaxf(v[i].w, 0.f);
          }
        }
        if (!CONV && masked) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (row0 + i >= m_end) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          split_store16(f32x4{v[0][e], v[1][e], v[2][e], v[3][e]}, S + swz16(4 * (cq + 32 * p) + e, chunk), BPL);
      }
    };
    producer_schedule<D, RS>(
        nks,
        [&](int s, auto setc) { load_a(s, setc); load_b(s, setc); __builtin_amdgcn_sched_barrier(0); },
        [&](int s, auto setc) {
          store_a(s, setc);
          if (s + D < nks) load_a(s + D, setc);
          __builtin_amdgcn_sched_barrier(0);
          store_b(s, setc);
          if (s + D < nks) load_b(s + D, setc);
          __builtin_amdgcn_sched_barrier(0);
        });
  } else {
    const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    consume_tile16<BM, BN, RS, kStr>(lds, nks, wm, wn, r, h, acc);
    const float inv = 1.f / (sa * sb);
    float* dst = partial ? partial + (size_t)slice * Cout * Cin : dW;
    const int taps = CONV ? Cin / Kc : 1;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = k0 + wn * (BN / 2) + j * 32 + r;
      if (CONV && col >= Cin) continue;
      const int tap = CONV ? col / Kc : 0, ci = col - tap * Kc;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = n0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (CONV && row >= Cout) continue;
          // pointwise and the convolutions' partial slices: [row][col] (lanes side by side); convolution without
          // scratch: torch's [co][ci][tap] - 36-byte strided atomics, several times the cost of the whole main loop
          if (partial) dst[(size_t)row * Cin + col] = acc[i][j][e] * inv;
          else atomicAdd(dst + ((size_t)row * Kc + ci) * taps + tap, acc[i][j][e] * inv);
        }
    }
  }
}

// dW[i] += partial[0][i] + partial[1][i] + ... (fixed order: bitwise reproducible)
__global__ void __launch_bounds__(256) wgrad_reduce_k(const float* __restrict__ partial, float* __restrict__ dW, int64_t n, int slices) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  float4 a = ld4(dW + i);
  for (int s = 0; s < slices; ++s) a = add4(a, ld4(partial + (size_t)s * n + i));
  st4(dW + i, a);
}

// dW[co][ci][tap] += partial[0][co][tap*Kc+ci] + partial[1][..] + ... (fixed order: bitwise reproducible); one thread per
// (co, tap, ci): the slice reads are coalesced, the one strided write per element is cheap
__global__ void __launch_bounds__(256) conv_wgrad_reduce_k(const float* __restrict__ partial, float* __restrict__ dW, int Cout, int Kc, int taps,
                                                            int slices) {
  const int64_t n = (int64_t)Cout * Kc * taps;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int ncols = Kc * taps;
  const int co = (int)(i / ncols), col = (int)(i - (int64_t)co * ncols), tap = col / Kc, ci = col - tap * Kc;
  float a = 0.f;
  for (int s = 0; s < slices; ++s) a += partial[(size_t)s * n + i];
  dW[((size_t)co * Kc + ci) * taps + tap] += a;
}

// The [M][K] x [Nout][K]^T shapes that run on these kernels (everything else: fp32 MFMA, pwconv.hip).
bool f16_gemm_shape(int K, int Nout) {
  return K >= 64 && K % 32 == 0 && ((Nout >= 256 && Nout % 256 == 0) || Nout == 128);
}
bool f16_wgrad_shape(int Cin, int Cout) {
  if (Cin < 128 || Cout < 128 || Cin % 128 || Cout % 128) return false;
  return Cin % 256 == 0 || Cout % 256 == 0 || (Cin == 128 && Cout == 128);
}
static int f16_wgrad_tiles(int Cin, int Cout) {
  if (Cin % 256 == 0) return (Cout / 128) * (Cin / 256);
  if (Cout % 256 == 0) return (Cout / 256) * (Cin / 128);
  return (Cout / 128) * (Cin / 128);
}

// slices of M for the weight gradient: one workgroup per CU, all of equal length
static void wgrad_slices(int64_t M, int tiles, int64_t& slices, int64_t& rows, int blocks = 256) {
  slices = blocks / tiles;
  if (slices < 1) slices = 1;
  const int64_t max_slices = ceil_div(M, 128);
  if (slices > max_slices) slices = max_slices;
  rows = ceil_div(ceil_div(M, slices), 32) * 32;
  slices = ceil_div(M, rows);
}

size_t f16_wgrad_partial_bytes(int64_t M, int Cin, int Cout) {
  if (!f16_wgrad_shape(Cin, Cout)) return 0;
  const int tiles = f16_wgrad_tiles(Cin, Cout);
  int64_t slices, rows;
  wgrad_slices(M, tiles, slices, rows);
  return (size_t)slices * Cin * Cout * sizeof(float);
}

template <typename T, typename TG>
bool launch_f16_wgrad(const TG* g, const T* y, const float* bn_pw, const T* ydw, const float* bn_dw, float* dw,
                      float* partial, int64_t M, int Cin, int Cout, hipStream_t st) {
  if (!f16_wgrad_shape(Cin, Cout)) return false;
  const bool wide = Cin % 256 == 0;  // 128 (Cout) x 256 (Cin) tiles, else 256 x 128, else (128 x 128 channels) one 128 x 128 tile
  const int tiles = f16_wgrad_tiles(Cin, Cout);
  int64_t slices, rows;
  wgrad_slices(M, tiles, slices, rows);
  const dim3 grid(tiles, (unsigned)slices);
  if (!wide && Cout % 256 != 0)
    hipLaunchKernelGGL((pw16_wgrad_k<128, 128, TTK_DW, T, TG>), grid, dim3(512), 0, st, g, y, bn_pw, ydw, bn_dw, dw, partial, M, Cin, Cout, rows, nullptr, ConvGeom{});
  else if (wide)
    hipLaunchKernelGGL((pw16_wgrad_k<128, 256, TTK_DW, T, TG>), grid, dim3(512), 0, st, g, y, bn_pw, ydw, bn_dw, dw, partial, M, Cin, Cout, rows, nullptr, ConvGeom{});
  else  // (one register set: two spill)
    hipLaunchKernelGGL((pw16_wgrad_k<256, 128, 1, T, TG>), grid, dim3(512), 0, st, g, y, bn_pw, ydw, bn_dw, dw, partial, M, Cin, Cout, rows, nullptr, ConvGeom{});
  if (partial) {
    const int64_t n = (int64_t)Cin * Cout;
    hipLaunchKernelGGL(wgrad_reduce_k, dim3((unsigned)ceil_div(n, 1024)), dim3(256), 0, st, partial, dw, n, (int)slices);
  }
  return true;
}

// ---- weight operand: |w| maximum of the layer (as ordered uint bits), then the two fp16 planes [K/32][rows][32] ----
__global__ void __launch_bounds__(256) w16_absmax_k(const float* __restrict__ w, int64_t n, unsigned* __restrict__ wmax) {
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0 && __float_as_uint(m) > __hip_atomic_load(wmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
    atomicMax(wmax, __float_as_uint(m));  // non-negative floats order like their bit patterns
}

// w[rows][K] fp32 -> two fp16 planes [K/32][rows][32] of w * pow2_scale(*wmax)
__global__ void w16_split_k(const float* __restrict__ w, uint16_t* __restrict__ q, const float* __restrict__ wmax, int rows, int K) {
  const int64_t n = (int64_t)rows * K;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float s = pow2_scale(*wmax);
  const int row = (int)(i / K), k = (int)(i - (int64_t)row * K);
  const int64_t o = ((int64_t)(k >> 5) * rows + row) * 32 + (k & 31);
  const float x = w[i] * s;
  const _Float16 hh = (_Float16)x;
  const _Float16 ll = (_Float16)(x - (float)hh);
  q[o] = __builtin_bit_cast(uint16_t, hh);
  q[n + o] = __builtin_bit_cast(uint16_t, ll);
}

// Returns true when the shape was handled here (and the kernels launched on `st`).  Bm != nullptr: raw weight rows
// [Nout][K] that are split into `planes` first (per-call form, unit tests); wmax: the layer's |w| maximum (a device
// float that the per-call form computes itself).
template <int MODE, typename T, typename TO>
bool launch_f16_gemm(const TO* A0, const T* A1, const float* bnA, const float* Bm, TO* out, const T* E0,
                     const float* bnE, float* part, int64_t M, int K, int Nout, void* planes, float* wmax, hipStream_t st) {
  constexpr int AM = MODE == SMODE_FWD ? AMODE_BNRELU : AMODE_BNGRAD, EM = MODE == SMODE_FWD ? EMODE_STATS : EMODE_MASK;
  if (!planes || !wmax || !f16_gemm_shape(K, Nout)) return false;
  uint16_t* Bq = reinterpret_cast<uint16_t*>(planes);
  const int64_t nw = (int64_t)Nout * K;
  if (Bm) {
    (void)hipMemsetAsync(wmax, 0, sizeof(float), st);
    hipLaunchKernelGGL(w16_absmax_k, dim3((unsigned)(nw / 1024 < 1 ? 1 : (nw / 1024 > 256 ? 256 : nw / 1024))), dim3(256), 0, st, Bm, nw,
                       reinterpret_cast<unsigned*>(wmax));
    hipLaunchKernelGGL(w16_split_k, dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, st, Bm, Bq, wmax, Nout, K);
  }
  if (Nout >= 256 && Nout % 256 == 0) {
    const unsigned tiles = (unsigned)(ceil_div(M, 128) * (Nout / 256));
    hipLaunchKernelGGL((pw16_k<128, 256, AM, EM, TTK_D, T, TO>), dim3(tiles), dim3(512), 0, st, A0, A1, bnA, Bq, wmax, out, E0, const_cast<float*>(bnE), part, M, K,
                       Nout, nullptr, ConvGeom{});
    return true;
  }
  if (Nout == 128) {
    const unsigned tiles = (unsigned)ceil_div(M, 256);
    constexpr int D = (MODE == SMODE_DGRAD || TTK_D > 2) ? 1 : TTK_D;  // eight A rows per thread: more sets spill
    hipLaunchKernelGGL((pw16_k<256, 128, AM, EM, D, T, TO>), dim3(tiles), dim3(512), 0, st, A0, A1, bnA, Bq, wmax, out, E0, const_cast<float*>(bnE), part, M, K,
                       Nout, nullptr, ConvGeom{});
    return true;
  }
  return false;
}

// dw[Cout][Kc][taps] += sum over output pixels of dy (x) gathered input activation (conv.hip)
// 64 output channels (ResNet layer1: 9 taps x 64 = 576 columns): 64x192 tiles = 3 taps each - no masked quarter as with
// 256-column tiles, and a tile's taps are the three kw of one kernel row, i.e. the same input rows one pixel apart
static int conv_wgrad_bn(int Cout, int ncols) { return Cout <= 64 && ncols % 192 == 0 && ncols % 256 != 0 ? 192 : 256; }
static void conv_wgrad_plan(int64_t M, int Cout, int ncols, int& tiles, int64_t& slices, int64_t& rows) {
  tiles = (int)(ceil_div(Cout, Cout <= 64 ? 64 : 128) * ceil_div(ncols, conv_wgrad_bn(Cout, ncols)));
  wgrad_slices(M, tiles, slices, rows, conv_wgrad_bn(Cout, ncols) == 192 ? 512 : 256);  // 64x192 tiles: two workgroups per CU
}
size_t conv_wgrad16_partial_bytes(int64_t M, int Cout, int ncols, int taps) {
  // 1x1: the atomics are coalesced ([co][ci] = the GEMM's layout) and measured faster than the fold - unless a fixed summation
  // order is asked for (TTK_DETERMINISTIC=1)
  if (taps == 1 && !deterministic_mode()) return 0;
  int tiles;
  int64_t slices, rows;
  conv_wgrad_plan(M, Cout, ncols, tiles, slices, rows);
  return (size_t)slices * Cout * ncols * sizeof(float);
}

bool launch_conv_wgrad16(const float* g, const float* y, const float* bn, const float* a_in, const float* a_bound, float* dw, float* partial,
                         int64_t M, int Cout, int taps, const ConvGeom& geo, hipStream_t st) {  // y == nullptr: g is dy
  if (geo.Kc % 4 != 0 || Cout % 4 != 0) return false;
  const int64_t nimg = M / ((int64_t)geo.Hg * geo.Wg);
  if (nimg * geo.Hs * geo.Ws * geo.Kc >= (int64_t)1 << 29) return false;  // 32-bit byte offsets into a_in (with the halo's slack)
  const int ncols = taps * geo.Kc;
  const bool narrow = Cout <= 64;
  if (taps == 1 && !deterministic_mode()) partial = nullptr;
  int tiles;
  int64_t slices, rows;
  conv_wgrad_plan(M, Cout, ncols, tiles, slices, rows);
  const dim3 grid(tiles, (unsigned)slices);
#ifndef TTK_DC
#define TTK_DC 1
#endif
  // y == nullptr: g holds dy as two fp16 planes [M][Cout] (h, then l)
  const float* yy = y ? y : reinterpret_cast<const float*>(reinterpret_cast<const uint16_t*>(g) + M * Cout);
#define TTK_WGRAD_LAUNCH(BM_, BN_, PLAIN_)                                                                                                      \
  hipLaunchKernelGGL((pw16_wgrad_k<BM_, BN_, (BM_ == 64 ? TTK_DC : 1), float, float, true, false, PLAIN_>), grid, dim3(512), 0, st, g, yy, bn, a_in, nullptr, dw, partial, M, ncols, \
                     Cout, rows, a_bound, geo)
  if (narrow && conv_wgrad_bn(Cout, ncols) == 192) { if (y) TTK_WGRAD_LAUNCH(64, 192, false); else TTK_WGRAD_LAUNCH(64, 192, true); }
  else if (narrow) { if (y) TTK_WGRAD_LAUNCH(64, 256, false); else TTK_WGRAD_LAUNCH(64, 256, true); }
  else             { if (y) TTK_WGRAD_LAUNCH(128, 256, false); else TTK_WGRAD_LAUNCH(128, 256, true); }
#undef TTK_WGRAD_LAUNCH
  if (partial)
    hipLaunchKernelGGL(conv_wgrad_reduce_k, dim3((unsigned)ceil_div((int64_t)Cout * ncols, 256)), dim3(256), 0, st, partial, dw, Cout, geo.Kc, taps,
                       (int)slices);
  return true;
}

// Implicit-GEMM convolution launches (conv.hip): Bq = two fp16 planes [K/32][Nout][32] (K = (tap, channel)) scaled by
// pow2_scale(*wmax).  Nout a multiple of 64, geo.Kc a multiple of 32.
bool launch_conv_gemm16(int amode, int emode, const float* A0, const float* A1, const float* bnA, const float* a_bound, const uint16_t* Bq,
                        const float* wmax, float* out, const float* E0, float* bnE, float* part, int64_t M, int K, int Nout,
                        const ConvGeom& geo, hipStream_t st) {
  ConvGeom gq = geo;
  gq.par = 0;
  // stride-2 data gradients without a masked epilogue (ResNet: every strided one): parity classes (conv_geom.h)
  const bool classes = geo.transposed && geo.stride == 2 && emode == EMODE_PLAIN && (geo.KW == 3 || geo.KW == 1);
  if (classes && geo.KW == 1)  // a strided 1x1 kernel reaches the (even, even) pixels only
    (void)hipMemsetAsync(out, 0, (size_t)M * Nout * sizeof(float), st);
  auto grid_of = [&](int bm, int bn) -> unsigned {
    if (!classes) return (unsigned)(ceil_div(M, bm) * (Nout / bn));
    gq.par = 1;
    gq.nimg = (int)(M / ((int64_t)geo.Hg * geo.Wg));
    int t = 0;
    for (int c = 0; c < 4; ++c) {
      const int ph = c < 2, pw = !(c & 1);
      const int64_t mc = (int64_t)gq.nimg * ((geo.Hg - ph + 1) >> 1) * ((geo.Wg - pw + 1) >> 1);
      gq.ctile[c] = t;
      if (geo.KW == 3 || c == 3) t += (int)(ceil_div(mc, bm) * (Nout / bn));
    }
    return (unsigned)t;
  };
#define TTK_CONV_LAUNCH(BM_, BN_, AM_, EM_)                                                                                        \
  do {                                                                                                                             \
    const unsigned grid_ = grid_of(BM_, BN_);                                                                                      \
    hipLaunchKernelGGL((pw16_k<BM_, BN_, AM_, EM_, 1, float, float, true>), dim3(grid_), dim3(512), 0, st, A0, A1, bnA, Bq, wmax, out, E0, bnE, \
                       part, M, K, Nout, a_bound, gq);                                                                             \
  } while (0)
#define TTK_CONV_TILES(AM_, EM_)                                   \
  do {                                                             \
    if (Nout % 256 == 0) TTK_CONV_LAUNCH(128, 256, AM_, EM_);      \
    else if (Nout % 128 == 0) TTK_CONV_LAUNCH(256, 128, AM_, EM_); \
    else if (narrow) TTK_CONV_LAUNCH(128, 64, AM_, EM_);           \
    else TTK_CONV_LAUNCH(256, 64, AM_, EM_);                       \
    return true;                                                   \
  } while (0)
  if (Nout % 64 != 0 || geo.Kc % 32 != 0) return false;
  static const bool narrow = !exp_env("TTK_CONV_WIDE64");  // 64 output channels: 128x64 tiles, two workgroups per CU (default) | 256x64
  if (amode == AMODE_PLAIN && emode == EMODE_STATS) TTK_CONV_TILES(AMODE_PLAIN, EMODE_STATS);
  if (amode == AMODE_BNGRAD && emode == EMODE_MASK) TTK_CONV_TILES(AMODE_BNGRAD, EMODE_MASK);
  if (amode == AMODE_BNGRAD && emode == EMODE_PLAIN) TTK_CONV_TILES(AMODE_BNGRAD, EMODE_PLAIN);
  if (amode == AMODE_PLAIN && emode == EMODE_MASK) TTK_CONV_TILES(AMODE_PLAIN, EMODE_MASK);  // data gradient of a materialised dy
  if (amode == AMODE_PLAIN && emode == EMODE_PLAIN) TTK_CONV_TILES(AMODE_PLAIN, EMODE_PLAIN);
  if (amode == AMODE_PLANES && emode == EMODE_MASK) TTK_CONV_TILES(AMODE_PLANES, EMODE_MASK);
  if (amode == AMODE_PLANES && emode == EMODE_PLAIN) TTK_CONV_TILES(AMODE_PLANES, EMODE_PLAIN);
  if (amode == AMODE_PLANES && emode == EMODE_STATS) TTK_CONV_TILES(AMODE_PLANES, EMODE_STATS);
#undef TTK_CONV_TILES
#undef TTK_CONV_LAUNCH
  return false;
}

#define TTK_INST(T_, TG_)                                                                                                              \
  template bool launch_f16_gemm<SMODE_FWD, T_, T_>(const T_*, const T_*, const float*, const float*, T_*, const T_*, const float*, float*, \
                                                   int64_t, int, int, void*, float*, hipStream_t);                                         \
  template bool launch_f16_gemm<SMODE_DGRAD, T_, TG_>(const TG_*, const T_*, const float*, const float*, TG_*, const T_*, const float*,    \
                                                      float*, int64_t, int, int, void*, float*, hipStream_t);                              \
  template bool launch_f16_wgrad<T_, TG_>(const TG_*, const T_*, const float*, const T_*, const float*, float*, float*, int64_t, int, int, \
                                          hipStream_t);
TTK_INST(float, float)
#undef TTK_INST

}  // namespace ttk
