// Pointwise 1x1 convolutions of the compute-bound layers (Cin >= 128, Cout >= 256 and the mirrored data
// gradients) as fp32 GEMMs on the bf16 matrix pipe.  Reference: DepthWiseBlock.conv_sep + bn_sep,
// backbones/mobilenet_v1.py:67-68,82-84.
//
// Arithmetic.  Every fp32 operand value x is cut EXACTLY into three bf16 pieces, x = h + m + l (each piece
// holds 8 significant bits: h = x truncated to bf16, m = (x-h) truncated, l = x-h-m, which then has at most
// 8 significant bits left).  A product a*b is the sum of nine piece products, each exact in fp32; the six
// of magnitude >= 2^-16 of the leading one (hl lh mm hm mh hh) are accumulated in fp32 by
// v_mfma_f32_32x32x16_bf16, the three below 2^-24 (ml lm ll) are dropped - the same order as the rounding
// of one fp32 fma.  Measured against an fp64 product the result is as close as an fp32 fmaf chain
// (tests/test_backbone_gpu.py, tools/exp/split_gemm_bench.hip); the bf16 pipe runs 16x the fp32 MFMA rate,
// so six products are 2.7x faster than v_mfma_f32_32x32x2_f32.
//
// Structure (one workgroup = 8 waves = one CU, block tile BM x BN = 128x256 or 256x128, k32 per step):
//   waves 4-7  PRODUCE: 16-byte global loads of both fp32 operands (128 B per row), the BatchNorm form of the
//              A operand (forward: relu(scale*(y-mean)+beta); data gradient: ga*(g-gmean)+gb*(y-mean)), the
//              3-way split (4 VALU + 1.5 v_perm per element) and ds_write_b64 into the LDS ring;
//   waves 0-3  CONSUME: ds_read_b128 fragments + MFMAs, wave tile (BM/2)x(BN/2), fragments of the streamed
//              operand double-buffered in registers so LDS latency hides under 12 MFMAs.
// The hardware places waves 0-3 and 4-7 of a workgroup on the four SIMDs in turn, so each SIMD runs one
// producer next to one consumer: VALU/LDS-write work and matrix work overlap by construction instead of
// alternating in lockstep phases.  One s_barrier per k32.
// LDS ring: 2 super-stages x 2 k16 stages; a stage holds 3 piece planes per operand as unpadded 32-byte rows
// (16 bf16), 16-byte chunk index XOR ((row>>3)&1): the 16 rows of a ds_read_b128 lane group fall on 16
// distinct 4-bank groups.  Stage stride 36864+64 B so the two k16 halves of a producer wave's ds_write_b64
// use different banks.
#include "ttk_common.h"
#include "conv_geom.h"

namespace ttk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

enum { SMODE_FWD = 0, SMODE_DGRAD = 1 };

constexpr int kStageBytes = 96 * (128 + 256);   // 3 planes x 32 B x (BM + BN) rows
constexpr int kStageStride = kStageBytes + 64;
constexpr int kRingBytes = 4 * kStageStride;

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 3) & 1)) << 4); }

__device__ __forceinline__ uint32_t pack_top16(float lo, float hi) {  // bf16(trunc lo) | bf16(trunc hi) << 16
  return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u);
}
#define TTK_RESID(x) ((x) - __uint_as_float(__float_as_uint(x) & 0xffff0000u))
// exact 3-way split of 4 consecutive-k values; writes the three 8-byte pieces at `dst` + piece*plane
__device__ __forceinline__ void split_store(f32x4 v, unsigned char* dst, int plane) {
  const float a1 = TTK_RESID(v.x), b1 = TTK_RESID(v.y), c1 = TTK_RESID(v.z), d1 = TTK_RESID(v.w);
  const float a2 = TTK_RESID(a1), b2 = TTK_RESID(b1), c2 = TTK_RESID(c1), d2 = TTK_RESID(d1);
  *reinterpret_cast<uint2*>(dst) = make_uint2(pack_top16(v.x, v.y), pack_top16(v.z, v.w));
  *reinterpret_cast<uint2*>(dst + plane) = make_uint2(pack_top16(a1, b1), pack_top16(c1, d1));
  *reinterpret_cast<uint2*>(dst + 2 * plane) = make_uint2(pack_top16(a2, b2), pack_top16(c2, d2));
}

// The consumer side of one block tile: `nks` super-stages (k32) of ds_read_b128 fragments + 6-product MFMAs into
// acc[TM][TN].  Executes exactly 1 + nks barriers (matching the producers).
template <int BM, int BN>
__device__ __forceinline__ void consume_tile(const unsigned char* lds, int nks, int wm, int wn, int r, int h,
                                             f32x16 (&acc)[BM / 64][BN / 64]) {
  constexpr int APL = BM * 32, BPL = BN * 32;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr bool HOLD_A = TM <= TN;  // hold the smaller fragment set in registers, stream the other
  constexpr int TH = HOLD_A ? TM : TN, TS = HOLD_A ? TN : TM;
  // byte offsets of this lane's fragments inside a stage
  int hold_off[TH], strm_off[TS];
#pragma unroll
  for (int x = 0; x < TH; ++x)
    hold_off[x] = HOLD_A ? swz_off(wm * (BM / 2) + x * 32 + r, h) : 3 * APL + swz_off(wn * (BN / 2) + x * 32 + r, h);
#pragma unroll
  for (int x = 0; x < TS; ++x)
    strm_off[x] = HOLD_A ? 3 * APL + swz_off(wn * (BN / 2) + x * 32 + r, h) : swz_off(wm * (BM / 2) + x * 32 + r, h);
  constexpr int HPL = HOLD_A ? APL : BPL, SPL = HOLD_A ? BPL : APL;

  bf16x8 hold[TH][3], hold_n[TH][3], strm[2][3];
  __syncthreads();  // super-stage 0 is in LDS
  for (int it = 0; it < nks; ++it) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      const unsigned char* S = lds + ((it & 1) * 2 + sub) * kStageStride;
      if (sub == 0) {  // first stage after the barrier: nothing could be prefetched across it
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
          for (int x = 0; x < TH; ++x) hold[x][p] = *reinterpret_cast<const bf16x8*>(S + p * HPL + hold_off[x]);
          strm[0][p] = *reinterpret_cast<const bf16x8*>(S + p * SPL + strm_off[0]);
        }
      }
#pragma unroll
      for (int x = 0; x < TS; ++x) {
        const int cur = (sub * TS + x) & 1;
        // prefetch the fragments of the next 12 (or 24) MFMAs
        if (x + 1 < TS) {
#pragma unroll
          for (int p = 0; p < 3; ++p) strm[cur ^ 1][p] = *reinterpret_cast<const bf16x8*>(S + p * SPL + strm_off[x + 1]);
        } else if (sub == 0) {
          const unsigned char* S2 = S + kStageStride;
#pragma unroll
          for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int y = 0; y < TH; ++y) hold_n[y][p] = *reinterpret_cast<const bf16x8*>(S2 + p * HPL + hold_off[y]);
            strm[cur ^ 1][p] = *reinterpret_cast<const bf16x8*>(S2 + p * SPL + strm_off[0]);
          }
        }
        // (Pinning [reads of the next group][12 MFMAs] with sched_barrier was measured 5-10 % SLOWER than hipcc's own
        // interleaving next to a producer wave on the same SIMD.)
        // six piece products, smallest first; (pa, pb) index the A and B pieces (0 = h, 1 = m, 2 = l)
#define TTK_PROD(pa, pb)                                                                                         \
  _Pragma("unroll") for (int y = 0; y < TH; ++y) {                                                               \
  if constexpr (HOLD_A)                                                                                        \
    acc[y][x] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hold[y][pa], strm[cur][pb], acc[y][x], 0, 0, 0);       \
  else                                                                                                         \
    acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(strm[cur][pa], hold[y][pb], acc[x][y], 0, 0, 0);       \
  }
        TTK_PROD(0, 2) TTK_PROD(2, 0) TTK_PROD(1, 1) TTK_PROD(0, 1) TTK_PROD(1, 0) TTK_PROD(0, 0)
#undef TTK_PROD
      }
      if (sub == 0) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int y = 0; y < TH; ++y) hold[y][p] = hold_n[y][p];
      }
    }
    __syncthreads();
  }
}

template <int BM, int BN, int AMODE, int EMODE, bool GATHER>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
pw_split_k(const float* __restrict__ A0, const float* __restrict__ A1, const float* __restrict__ bnA,
           const uint16_t* __restrict__ Bq, float* __restrict__ out, const float* __restrict__ E0,
           const float* __restrict__ bnE, float* __restrict__ part, int64_t M, int K, int Nout, ConvGeom geo) {
  static_assert((BM == 128 && BN == 256) || (BM == 256 && BN == 128) || (BM == 256 && BN == 64), "tile shapes");
  constexpr int APL = BM * 32, BPL = BN * 32;  // bytes of one piece plane
  constexpr int TM = BM / 64, TN = BN / 64;    // 32x32 tiles of a consumer wave (wave tile (BM/2) x (BN/2))
  constexpr int LDC = BN + 4;
  constexpr int QN = BN / 4, RG = 512 / QN, HALVES = BM / 128, RGH = RG / HALVES;
  constexpr int kEpiBytes = BM * LDC * 4 + RG * 2 * BN * 4;
  constexpr int kSmemBytes = kRingBytes > kEpiBytes ? kRingBytes : kEpiBytes;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kSmemBytes];

  const int tid = threadIdx.x;
  // XCD-aware tile order (see pwconv.hip): every XCD gets a contiguous range of tiles.
  const unsigned G = gridDim.x, Lid = blockIdx.x, NB = Nout / BN;
  const unsigned xq = G / 8, xr = G % 8, xcd = Lid % 8;
  const unsigned tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + Lid / 8;
  const unsigned bx = tile % NB, by = tile / NB;
  const int64_t m0 = (int64_t)by * BM;
  const int n0 = bx * BN;
  const int nks = K / 32;
  const bool producer = __builtin_amdgcn_readfirstlane(tid) >= 256;

  if (producer) {
    __builtin_amdgcn_s_setprio(3);  // the producers are the critical path: let them issue ahead of the MFMA waves (-0.35 % step time)
    // ------------------------------------------------------------------ producer waves
    const int pt = tid - 256;
    const int row0 = pt >> 3, kq8 = pt & 7;  // 32 rows per pass; 8 lanes x 16 B = one 128-byte row segment
    const int sub = kq8 >> 2, chunk = (kq8 >> 1) & 1, o8 = (kq8 & 1) * 8;
    constexpr int AP = BM / 32, BP = BN / 32;
    f32x4 ra0[AP], ra1[AMODE == AMODE_BNGRAD ? AP : 1], q0, q1, q2, q3;
    constexpr int BI = BN / 64;  // B rows per thread and piece plane: 64 rows x 4 chunks of 16 B (8 k) per pass
    u32x4 rb[1][3][BI];
    const int Kc = GATHER ? geo.Kc : K;   // channels per tap (= K without taps)
    const int kpt = Kc / 32;              // k32 steps per tap
    int64_t arow[GATHER ? 1 : AP];        // plain GEMM: element offset of each of this thread's rows
    int gbase[GATHER ? AP : 1], gh[GATHER ? AP : 1], gw[GATHER ? AP : 1];  // gather: image base pixel, grid coordinates
    unsigned vmask = 0xffffffffu;         // rows whose current tap falls inside the source tensor
    if constexpr (!GATHER) {
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        int64_t row = m0 + row0 + 32 * i;
        arow[i] = (row < M ? row : M - 1) * (int64_t)K + kq8 * 4;  // clamp: rows past M are computed but never stored
      }
    } else {
      const int hw = geo.Hg * geo.Wg;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        const int64_t row = m0 + row0 + 32 * i;
        const int r = (int)(row < M ? row : M - 1);
        const int n = r / hw, rem = r - n * hw;
        gh[i] = rem / geo.Wg;
        gw[i] = rem - gh[i] * geo.Wg;
        gbase[i] = row < M ? n * geo.Hs * geo.Ws : -1;
      }
    }
    // B arrives already split (launch_*: split_weights_k / conv_weight_repack_k / pw_prepare_weights_k): three bf16
    // planes [K/32][Nout][32] - the 64 bytes a row contributes to one k32 step are contiguous and rows follow each
    // other, so every wave-wide 16-byte load reads ONE contiguous KB (whole cache lines; a [Nout][K] plane gave each
    // load sixteen half-used lines whose other halves had left the L1 by the next step).
    const int brow = pt >> 2, bc4 = pt & 3;
    const uint16_t* bp = Bq + (int64_t)(n0 + brow) * 32 + bc4 * 8;
    const int64_t bplane = (int64_t)K * Nout;
    unsigned char* wbase_b = lds + (bc4 >> 1) * kStageStride + 3 * APL;
    const float* cp = bnA ? bnA + kq8 * 4 : nullptr;
    unsigned char* wbase = lds + sub * kStageStride + o8;

    auto load_a = [&](int ks) {
      const int tap = ks / kpt, kc0 = (ks - tap * kpt) * 32;
      if constexpr (!GATHER) {
#pragma unroll
        for (int i = 0; i < AP; ++i) {
          ra0[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A0 + arow[i] + kc0));  // streamed: +5 % on the data gradient
          if constexpr (AMODE == AMODE_BNGRAD) ra1[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(A1 + arow[i] + kc0));
        }
      } else {
        const int kh = tap / geo.KW, kw = tap - kh * geo.KW;
        vmask = 0u;
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
          const int64_t off = ok ? ((int64_t)(gbase[i] + sh * geo.Ws + sw) * Kc + kc0 + kq8 * 4) : (int64_t)(kq8 * 4);
          vmask |= (ok ? 1u : 0u) << i;
          ra0[i] = *reinterpret_cast<const f32x4*>(A0 + off);
          if constexpr (AMODE == AMODE_BNGRAD) ra1[i] = *reinterpret_cast<const f32x4*>(A1 + off);
        }
      }
      if constexpr (AMODE == AMODE_BNRELU) {
        q0 = *reinterpret_cast<const f32x4*>(cp + TTK_BN_SCALE * Kc + kc0);
        q1 = *reinterpret_cast<const f32x4*>(cp + TTK_BN_MEAN * Kc + kc0);
        q2 = *reinterpret_cast<const f32x4*>(cp + TTK_BN_BETA * Kc + kc0);
      } else if constexpr (AMODE == AMODE_BNGRAD) {
        q0 = *reinterpret_cast<const f32x4*>(cp + TTK_BN_GA * Kc + kc0);
        q1 = *reinterpret_cast<const f32x4*>(cp + TTK_BN_GMEAN * Kc + kc0);
        q2 = *reinterpret_cast<const f32x4*>(cp + TTK_BN_GB * Kc + kc0);
        q3 = *reinterpret_cast<const f32x4*>(cp + TTK_BN_MEAN * Kc + kc0);
      }
    };
    auto load_b = [&](int ks, int set) {
      const uint16_t* b = bp + (int64_t)ks * Nout * 32;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < BI; ++i) rb[set][p][i] = *reinterpret_cast<const u32x4*>(b + p * bplane + 64 * 32 * i);
    };
    auto store_a = [&](int ks) {
      unsigned char* S = wbase + (ks & 1) * 2 * kStageStride;
#pragma unroll
      for (int i = 0; i < AP; ++i) {
        f32x4 v;
        if constexpr (AMODE == AMODE_BNRELU) {
          v = q0 * (ra0[i] - q1) + q2;
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        } else if constexpr (AMODE == AMODE_BNGRAD) {
          v = q0 * (ra0[i] - q1) + q2 * (ra1[i] - q3);
        } else {
          v = ra0[i];
        }
        if constexpr (GATHER)
          if (!((vmask >> i) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};  // zero padding / taps that miss the stride grid
        split_store(v, S + swz_off(row0 + 32 * i, chunk), APL);
      }
    };
    auto store_b = [&](int ks, int set) {  // no arithmetic: 16-byte chunks (8 k of one piece) straight into the ring
      unsigned char* S = wbase_b + (ks & 1) * 2 * kStageStride;
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < BI; ++i) *reinterpret_cast<u32x4*>(S + p * BPL + swz_off(brow + 64 * i, bc4 & 1)) = rb[set][p][i];
    };

    // Cycle stamps of this loop (s_memtime around each phase, 128x256 forward tile): ~3300 cycles of MFMA issue per k32 in
    // the consumers, which then waited 900-2000 cycles at the barrier for the producers; the producers' time was mostly
    // spent waiting for loads issued only ~1200 cycles before their use.
    // One register set per operand is enough when every load is issued right after the previous contents of its
    // registers were consumed: B is consumed first in a step and reloaded at once, then A - each load has a whole
    // step (~4000 cycles) to land, with 16 loads in flight per thread (more - a second set per operand - overflowed
    // the CU's vector-memory queue and stalled the load ISSUE for thousands of cycles).
    auto step = [&](int it) {  // fills super-stage it+1
      if (it + 1 < nks) {
        store_b(it + 1, 0);
        if (it + 2 < nks) load_b(it + 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        store_a(it + 1);
        if (it + 2 < nks) load_a(it + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    load_b(0, 0);
    load_a(0);
    __builtin_amdgcn_sched_barrier(0);
    store_b(0, 0);
    if (nks > 1) load_b(1, 0);
    __builtin_amdgcn_sched_barrier(0);
    store_a(0);
    if (nks > 1) load_a(1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();  // super-stage 0 is in LDS
    for (int it = 0; it < nks; ++it) {
      step(it);
      __syncthreads();
    }
  } else {
    // ------------------------------------------------------------------ consumer waves
    const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    consume_tile<BM, BN>(lds, nks, wm, wn, r, h, acc);
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
  __syncthreads();

  // ------------------------------------------------------------------ epilogue, all 8 waves
  // Row-wise pass over the C image: 16-byte stores (a wave writes 1 KB row segments), the ReLU mask of the data
  // gradient, and the BatchNorm partial sums of this tile's 128-row halves.
  const float* Cs = reinterpret_cast<const float*>(lds);
  float* red = reinterpret_cast<float*>(lds + BM * LDC * 4);  // [RG][2][BN]
  const int c4 = tid % QN, rg = tid / QN, half = rg / RGH, rr = rg % RGH;
  const int col = n0 + 4 * c4;
  float4 esc = f4(0.f), emean = f4(0.f), ebeta = f4(0.f);
  if constexpr (EMODE == EMODE_MASK) {
    esc = ld4(bnE + TTK_BN_SCALE * Nout + col); emean = ld4(bnE + TTK_BN_MEAN * Nout + col); ebeta = ld4(bnE + TTK_BN_BETA * Nout + col);
  }
  if constexpr (EMODE == EMODE_STATS) {
    if (bnE) emean = ld4(bnE + col);  // forward: bnE is the statistics pivot [Nout]
  }
  float4 s1 = f4(0.f), s2 = f4(0.f);
#pragma unroll 4
  for (int i = 0; i < 128 / RGH; ++i) {
    const int row = half * 128 + rr + RGH * i;
    const int64_t grow = m0 + row;
    if (grow >= M) break;
    float4 v = ld4(Cs + row * LDC + 4 * c4);
    const size_t o = (size_t)grow * Nout + col;
    if constexpr (EMODE == EMODE_PLAIN) {
      st4(out + o, v);
    } else if constexpr (EMODE == EMODE_STATS) {
      st4(out + o, v);
      v = sub4(v, emean);
      s1 = add4(s1, v);
      s2 = fma4(v, v, s2);
    } else {
      const float4 yc = sub4(ld4(E0 + o), emean);
      v = mask4(v, fma4(esc, yc, ebeta));
      st4(out + o, v);
      s1 = add4(s1, v);
      s2 = fma4(v, yc, s2);
    }
  }
  if (EMODE != EMODE_PLAIN && part) {
    st4(red + (rg * 2 + 0) * BN + 4 * c4, s1);
    st4(red + (rg * 2 + 1) * BN + 4 * c4, s2);
    __syncthreads();
    for (int i = tid; i < HALVES * 2 * BN; i += 512) {
      const int hf = i / (2 * BN), which = (i / BN) & 1, c = i % BN;
      float a = 0.f;
      for (int q = 0; q < RGH; ++q) a += red[((hf * RGH + q) * 2 + which) * BN + c];  // fixed order: reproducible
      const int64_t prow = (int64_t)by * HALVES + hf;
      if (prow * 128 < M) part[(size_t)prow * 2 * Nout + (size_t)which * Nout + n0 + c] = a;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient  dW[co][ci] += sum_m dy[m][co] * a[m][ci]  on the same consumer pipeline: the contraction runs
// over the rows m, so the MFMA fragments need 8 CONSECUTIVE m of one channel.  Each producer thread loads a
// 4 (rows) x 4 (channels) block - four 16-byte loads, lanes of a row group side by side in the channel
// direction (128-byte segments) - and the transposition is free: register e of the four rows IS the 4
// consecutive m of channel e, which is split and written as one 8-byte piece into the [channel][m] image.
// grid.x = dW tiles, grid.y = slices of M (one fp32 atomicAdd per output element and slice).
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, bool CONV>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
pw_split_wgrad_k(const float* __restrict__ G, const float* __restrict__ Y, const float* __restrict__ bn_pw,
                 const float* __restrict__ X, const float* __restrict__ bn_x, float* __restrict__ dW, int64_t M, int Ncols,
                 int Cout, int64_t rows_per_slice, ConvGeom geo) {
  // Pointwise (CONV = false): columns = input channels, X = raw depthwise output (BatchNorm+ReLU applied on load).
  // Convolution (CONV = true): columns = (tap, input channel), X = the materialised input activation [B][Hs][Ws][Kc];
  // the rows m enumerate OUTPUT pixels (Hg x Wg per image) and column (tap, ci) reads the input pixel the tap points
  // at (zero outside).  Partial tiles (Cout < BM, Ncols % BN != 0) are masked.
  static_assert(BM + BN == 384 && (BM == 128 || BM == 256), "128x256 or 256x128");
  constexpr int APL = BM * 32, BPL = BN * 32;
  constexpr int TM = BM / 64, TN = BN / 64;
  __shared__ __attribute__((aligned(16))) unsigned char lds[kRingBytes];

  const int tid = threadIdx.x;
  // XCD-aware order: workgroup ids go round-robin over the 8 XCDs; give each XCD whole slices (all dW tiles of a
  // slice run side by side on ONE L2, so the slice's operand rows are fetched from HBM once, not once per tile).
  const unsigned T = gridDim.x, NG = T * gridDim.y, Lid = blockIdx.y * T + blockIdx.x;
  const unsigned xq = NG / 8, xr = NG % 8, xcd = Lid % 8;
  const unsigned logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + Lid / 8;
  const unsigned tile = logical % T, slice = logical / T;
  const int tiles_k = (Ncols + BN - 1) / BN;
  const int n0 = (tile / tiles_k) * BM, k0 = (tile % tiles_k) * BN;
  const int Kc = CONV ? geo.Kc : Ncols;  // row length of X
  const int64_t m_begin = (int64_t)slice * rows_per_slice;
  const int64_t m_end = (m_begin + rows_per_slice < M) ? m_begin + rows_per_slice : M;
  if (m_begin >= m_end) return;  // uniform over the block, before any barrier
  const int nks = (int)((m_end - m_begin + 31) / 32);
  const bool producer = __builtin_amdgcn_readfirstlane(tid) >= 256;

  if (producer) {
    __builtin_amdgcn_s_setprio(3);
    const int pt = tid - 256;
    const int mb = pt & 7, cq = pt >> 3;  // 8 row blocks of 4 rows x 32 channel quads per pass
    const int sub = mb >> 2, chunk = (mb >> 1) & 1, o8 = (mb & 1) * 8;
    constexpr int AP = BM / 128, BP = BN / 128;
    f32x4 rg[AP][4], ry[AP][4], rx[BP][4];
    f32x4 ga[AP], gb[AP], gmean[AP], ymean[AP], sc[BP], mu[BP], be[BP];
    int ca[AP], cb[BP], kh[BP], kw[BP];  // channel of the A rows; input channel and tap of the B columns
    bool va[AP], vb[BP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      const int c = n0 + 4 * (cq + 32 * p);
      va[p] = c < Cout;
      ca[p] = va[p] ? c : 0;
      ga[p] = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_GA * Cout + ca[p]);
      gb[p] = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_GB * Cout + ca[p]);
      gmean[p] = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_GMEAN * Cout + ca[p]);
      ymean[p] = *reinterpret_cast<const f32x4*>(bn_pw + TTK_BN_MEAN * Cout + ca[p]);
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) {
      const int c = k0 + 4 * (cq + 32 * p);
      vb[p] = c < Ncols;
      const int cc = vb[p] ? c : 0;
      const int tap = CONV ? cc / Kc : 0;
      cb[p] = cc - tap * Kc;
      kh[p] = CONV ? tap / geo.KW : 0;
      kw[p] = CONV ? tap - kh[p] * geo.KW : 0;
      if constexpr (!CONV) {
        sc[p] = *reinterpret_cast<const f32x4*>(bn_x + TTK_BN_SCALE * Kc + cb[p]);
        mu[p] = *reinterpret_cast<const f32x4*>(bn_x + TTK_BN_MEAN * Kc + cb[p]);
        be[p] = *reinterpret_cast<const f32x4*>(bn_x + TTK_BN_BETA * Kc + cb[p]);
      }
    }
    unsigned bmask = 0u;  // (pass, row) pairs of the in-flight B loads that hit the source tensor
    unsigned char* wbase = lds + sub * kStageStride + o8;

    // Steps whose 32 rows all lie inside the slice (all but possibly the last one) take a path without row clamps and
    // zero fills, and the pointwise kernel's tiles are always full (launch_split_wgrad): the producers are the critical
    // path of this kernel (VALU issue, ~850 instructions per step before this split), so every select counts.
    const int nfull = (int)((m_end - m_begin) / 32);
    const float* gp[AP];
    const float* yp[AP];
    const float* xp[BP];
#pragma unroll
    for (int p = 0; p < AP; ++p) {
      gp[p] = G + (m_begin + 4 * mb) * Cout + ca[p];
      yp[p] = Y + (m_begin + 4 * mb) * Cout + ca[p];
    }
#pragma unroll
    for (int p = 0; p < BP; ++p) xp[p] = X + (m_begin + 4 * mb) * Kc + cb[p];

    auto load_a = [&](int ks) {
      if (ks < nfull) {
        const int64_t base = (int64_t)ks * 32 * Cout;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int p = 0; p < AP; ++p) {
            rg[p][i] = *reinterpret_cast<const f32x4*>(gp[p] + base + (int64_t)i * Cout);
            ry[p][i] = *reinterpret_cast<const f32x4*>(yp[p] + base + (int64_t)i * Cout);
          }
        return;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        int64_t row = m_begin + (int64_t)ks * 32 + 4 * mb + i;
        row = row < m_end ? row : m_end - 1;
#pragma unroll
        for (int p = 0; p < AP; ++p) {
          rg[p][i] = *reinterpret_cast<const f32x4*>(G + row * Cout + ca[p]);
          ry[p][i] = *reinterpret_cast<const f32x4*>(Y + row * Cout + ca[p]);
        }
      }
    };
    auto load_b = [&](int ks) {
      const int64_t r0 = m_begin + (int64_t)ks * 32 + 4 * mb;
      if constexpr (!CONV) {
        if (ks < nfull) {
          const int64_t base = (int64_t)ks * 32 * Kc;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p = 0; p < BP; ++p) rx[p][i] = *reinterpret_cast<const f32x4*>(xp[p] + base + (int64_t)i * Kc);
          return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          int64_t row = r0 + i;
          row = row < m_end ? row : m_end - 1;
#pragma unroll
          for (int p = 0; p < BP; ++p) rx[p][i] = *reinterpret_cast<const f32x4*>(X + row * Kc + cb[p]);
        }
      } else {
        // (n, ho, wo) of the first of the four consecutive output pixels by division, the others by carry
        const int hw = geo.Hg * geo.Wg;
        const int rr = (int)(r0 < m_end ? r0 : m_end - 1);
        int n = rr / hw, rem = rr - n * hw, ho = rem / geo.Wg, wo = rem - ho * geo.Wg;
        bmask = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool rowok = r0 + i < m_end;
#pragma unroll
          for (int p = 0; p < BP; ++p) {
            const int hi = ho * geo.stride - geo.pad + kh[p], wi = wo * geo.stride - geo.pad + kw[p];
            const bool ok = rowok && vb[p] && (unsigned)hi < (unsigned)geo.Hs && (unsigned)wi < (unsigned)geo.Ws;
            const int64_t off = ok ? ((int64_t)((n * geo.Hs + hi) * geo.Ws + wi) * Kc + cb[p]) : (int64_t)cb[p];
            bmask |= (ok ? 1u : 0u) << (4 * p + i);
            rx[p][i] = *reinterpret_cast<const f32x4*>(X + off);
          }
          if (++wo == geo.Wg) {
            wo = 0;
            if (++ho == geo.Hg) { ho = 0; ++n; }
          }
        }
      }
    };
    auto store_a = [&](int ks) {
      unsigned char* S = wbase + (ks & 1) * 2 * kStageStride;
      const int64_t row0 = m_begin + (int64_t)ks * 32 + 4 * mb;
      const bool masked = CONV || ks >= nfull;  // uniform
#pragma unroll
      for (int p = 0; p < AP; ++p) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ga[p] * (rg[p][i] - gmean[p]) + gb[p] * (ry[p][i] - ymean[p]);
        if (masked) {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (row0 + i >= m_end || !va[p]) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};  // rows past the slice contribute nothing
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          split_store(f32x4{v[0][e], v[1][e], v[2][e], v[3][e]}, S + swz_off(4 * (cq + 32 * p) + e, chunk), APL);
      }
    };
    auto store_b = [&](int ks) {
      unsigned char* S = wbase + (ks & 1) * 2 * kStageStride + 3 * APL;
      const int64_t row0 = m_begin + (int64_t)ks * 32 + 4 * mb;
      const bool masked = ks >= nfull;  // uniform (pointwise)
#pragma unroll
      for (int p = 0; p < BP; ++p) {
        f32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (!CONV) {
            v[i] = sc[p] * (rx[p][i] - mu[p]) + be[p];
            v[i].x = fmaxf(v[i].x, 0.f); v[i].y = fmaxf(v[i].y, 0.f); v[i].z = fmaxf(v[i].z, 0.f); v[i].w = fmaxf(v[i].w, 0.f);
          } else {
            v[i] = ((bmask >> (4 * p + i)) & 1u) ? rx[p][i] : f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
        if constexpr (!CONV) {
          if (masked) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (row0 + i >= m_end || !vb[p]) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
          split_store(f32x4{v[0][e], v[1][e], v[2][e], v[3][e]}, S + swz_off(4 * (cq + 32 * p) + e, chunk), BPL);
      }
    };

    load_a(0);
    load_b(0);
    store_a(0);
    if (nks > 1) load_a(1);
    __builtin_amdgcn_sched_barrier(0);
    store_b(0);
    if (nks > 1) load_b(1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    for (int it = 0; it < nks; ++it) {
      if (it + 1 < nks) {
        store_a(it + 1);
        if (it + 2 < nks) load_a(it + 2);
        __builtin_amdgcn_sched_barrier(0);
        store_b(it + 1);
        if (it + 2 < nks) load_b(it + 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
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
    consume_tile<BM, BN>(lds, nks, wm, wn, r, h, acc);
    const int taps = CONV ? Ncols / Kc : 1;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = k0 + wn * (BN / 2) + j * 32 + r;
      if (col >= Ncols) continue;
      const int tap = CONV ? col / Kc : 0, ci = col - tap * Kc;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = n0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (row < Cout) atomicAdd(dW + ((size_t)row * Kc + ci) * taps + tap, acc[i][j][e]);  // dw[co][ci][tap]
        }
    }
  }
}

bool launch_split_wgrad(const float* g, const float* y, const float* bn_pw, const float* ydw, const float* bn_dw, float* dw,
                        int64_t M, int Cin, int Cout, hipStream_t st) {
  if (Cin < 128 || Cout < 128 || Cin % 128 || Cout % 128 || (int64_t)Cin * Cout < 128 * 256) return false;
  const bool wide = Cin % 256 == 0;  // 128 (Cout) x 256 (Cin) tiles, else 256 x 128
  if (!wide && Cout % 256) return false;
  const int tiles = wide ? (Cout / 128) * (Cin / 256) : (Cout / 256) * (Cin / 128);
  int64_t slices = 256 / tiles;  // one workgroup per CU, all of equal length
  if (slices < 1) slices = 1;
  const int64_t max_slices = ceil_div(M, 128);
  if (slices > max_slices) slices = max_slices;
  const int64_t rows = ceil_div(ceil_div(M, slices), 32) * 32;
  slices = ceil_div(M, rows);
  const dim3 grid(tiles, (unsigned)slices);
  if (wide)
    hipLaunchKernelGGL((pw_split_wgrad_k<128, 256, false>), grid, dim3(512), 0, st, g, y, bn_pw, ydw, bn_dw, dw, M, Cin, Cout, rows, ConvGeom{});
  else
    hipLaunchKernelGGL((pw_split_wgrad_k<256, 128, false>), grid, dim3(512), 0, st, g, y, bn_pw, ydw, bn_dw, dw, M, Cin, Cout, rows, ConvGeom{});
  return true;
}

// dw[Cout][Cin][taps] += sum over output pixels of dy (x) gathered input activation (conv.hip)
bool launch_conv_wgrad(const float* g, const float* y, const float* bn, const float* a_in, float* dw, int64_t M, int Cout,
                       int taps, const ConvGeom& geo, hipStream_t st) {
  if (geo.Kc % 4 != 0 || Cout % 4 != 0) return false;
  const int ncols = taps * geo.Kc;
  const int tiles = (int)(ceil_div(Cout, 128) * ceil_div(ncols, 256));
  int64_t slices = 256 / tiles;
  if (slices < 1) slices = 1;
  const int64_t max_slices = ceil_div(M, 128);
  if (slices > max_slices) slices = max_slices;
  const int64_t rows = ceil_div(ceil_div(M, slices), 32) * 32;
  slices = ceil_div(M, rows);
  hipLaunchKernelGGL((pw_split_wgrad_k<128, 256, true>), dim3(tiles, (unsigned)slices), dim3(512), 0, st, g, y, bn, a_in, nullptr, dw,
                     M, ncols, Cout, rows, geo);
  return true;
}

// w[n] fp32 -> q[3][n] bf16 pieces (same element order): the exact 3-way split of split_store, done once per call
// for the weight operand so that the GEMM producers move it without arithmetic.
// w[Nout][K] fp32 -> three bf16 planes [K/32][Nout][32] (the layout pw_split_k's producers read)
__global__ void split_weights_k(const float* __restrict__ w, uint16_t* __restrict__ q, int Nout, int K) {
  const int64_t n = (int64_t)Nout * K;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int row = (int)(i / K), k = (int)(i - (int64_t)row * K);
  const int64_t o = ((int64_t)(k >> 5) * Nout + row) * 32 + (k & 31);
  const float x = w[i];
  const float r1 = TTK_RESID(x), r2 = TTK_RESID(r1);
  q[o] = (uint16_t)(__float_as_uint(x) >> 16);
  q[n + o] = (uint16_t)(__float_as_uint(r1) >> 16);
  q[2 * n + o] = (uint16_t)(__float_as_uint(r2) >> 16);
}


// The [M][K] x [Nout][K]^T shapes that run on the split kernels (everything else: fp32 MFMA, pwconv.hip).
bool split_gemm_shape(int K, int Nout) {
  return K >= 128 && K % 32 == 0 && ((Nout >= 256 && Nout % 256 == 0) || Nout == 128);
}

// Returns true when the shape was handled here (and the kernel launched on `st`).  Bm == nullptr: wsplit already holds
// the three planes (ttk_pwconv_prepare_weights).
template <int MODE>
bool launch_split_gemm(const float* A0, const float* A1, const float* bnA, const float* Bm, float* out, const float* E0,
                       const float* bnE, float* part, int64_t M, int K, int Nout, void* wsplit, hipStream_t st) {
  constexpr int AM = MODE == SMODE_FWD ? AMODE_BNRELU : AMODE_BNGRAD, EM = MODE == SMODE_FWD ? EMODE_STATS : EMODE_MASK;
  if (!wsplit || !split_gemm_shape(K, Nout)) return false;
  const ConvGeom none{};
  uint16_t* Bq = reinterpret_cast<uint16_t*>(wsplit);
  const int64_t nw = (int64_t)Nout * K;
  if (Bm) hipLaunchKernelGGL(split_weights_k, dim3((unsigned)ceil_div(nw, 256)), dim3(256), 0, st, Bm, Bq, Nout, K);
  if (Nout >= 256 && Nout % 256 == 0) {
    const unsigned tiles = (unsigned)(ceil_div(M, 128) * (Nout / 256));
    hipLaunchKernelGGL((pw_split_k<128, 256, AM, EM, false>), dim3(tiles), dim3(512), 0, st, A0, A1, bnA, Bq, out, E0, bnE, part, M, K,
                       Nout, none);
    return true;
  }
  if (Nout == 128 && K >= 128) {
    const unsigned tiles = (unsigned)ceil_div(M, 256);
    hipLaunchKernelGGL((pw_split_k<256, 128, AM, EM, false>), dim3(tiles), dim3(512), 0, st, A0, A1, bnA, Bq, out, E0, bnE, part, M, K,
                       Nout, none);
    return true;
  }
  return false;
}

template bool launch_split_gemm<SMODE_FWD>(const float*, const float*, const float*, const float*, float*, const float*,
                                           const float*, float*, int64_t, int, int, void*, hipStream_t);
template bool launch_split_gemm<SMODE_DGRAD>(const float*, const float*, const float*, const float*, float*, const float*,
                                             const float*, float*, int64_t, int, int, void*, hipStream_t);

// Implicit-GEMM convolution launches (conv.hip).  amode/emode: AMODE_* / EMODE_*.  Nout must be a multiple of 64;
// geo.Kc a multiple of 32.  K = taps * geo.Kc.
bool launch_conv_gemm(int amode, int emode, const float* A0, const float* A1, const float* bnA, const uint16_t* Bm, float* out,
                      const float* E0, const float* bnE, float* part, int64_t M, int K, int Nout, const ConvGeom& geo,
                      hipStream_t st) {
#define TTK_CONV_LAUNCH(BM_, BN_, AM_, EM_)                                                                           \
  hipLaunchKernelGGL((pw_split_k<BM_, BN_, AM_, EM_, true>), dim3((unsigned)(ceil_div(M, BM_) * (Nout / BN_))), dim3(512), 0, st, \
                     A0, A1, bnA, Bm, out, E0, bnE, part, M, K, Nout, geo)
#define TTK_CONV_TILES(AM_, EM_)                                  \
  do {                                                            \
    if (Nout % 256 == 0) TTK_CONV_LAUNCH(128, 256, AM_, EM_);     \
    else if (Nout % 128 == 0) TTK_CONV_LAUNCH(256, 128, AM_, EM_); \
    else TTK_CONV_LAUNCH(256, 64, AM_, EM_);                      \
    return true;                                                  \
  } while (0)
  if (Nout % 64 != 0 || geo.Kc % 32 != 0) return false;
  if (amode == AMODE_PLAIN && emode == EMODE_STATS) TTK_CONV_TILES(AMODE_PLAIN, EMODE_STATS);
  if (amode == AMODE_BNGRAD && emode == EMODE_MASK) TTK_CONV_TILES(AMODE_BNGRAD, EMODE_MASK);
  if (amode == AMODE_BNGRAD && emode == EMODE_PLAIN) TTK_CONV_TILES(AMODE_BNGRAD, EMODE_PLAIN);
#undef TTK_CONV_TILES
#undef TTK_CONV_LAUNCH
  return false;
}

}  // namespace ttk
