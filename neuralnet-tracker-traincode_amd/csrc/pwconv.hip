// Pointwise 1x1 convolutions as GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32,
// bit-for-bit an fmaf chain, 64 cycles per issue per SIMD).  Reference: DepthWiseBlock.conv_sep +
// bn_sep, backbones/mobilenet_v1.py:67-68,82-84; 95.7 % of the network's MACs.
//
//   forward    Y[M][N]  = relu(scale*Ydw + shift)[M][K] . W[N][K]^T          (+ sum(y), sum(y^2) per column)
//   data grad  Gdw[M][K] = (dY[M][N] . W[N][K]) * [bn_dw(Ydw) > 0]            (+ sum(g), sum(g*ydw) per column)
//   weight grad dW[N][K] += dY[M][N]^T . relu(scale*Ydw + shift)[M][K]
// with dY = cA*G + cB*Y + cC formed while loading (BatchNorm backward folded into three
// per-channel coefficients).  M = B*H*W rows of channels-last activations, so both operands of
// forward/data-grad are row-major with the contraction index contiguous: 16-byte global loads,
// ds_write_b128 into a +4-float padded LDS image, conflict-free ds_read_b128 fragments
// (row stride 36 floats: 16 consecutive rows hit 16 distinct 4-bank groups).
//
// A ds_read_b128 hands each lane 4 consecutive k of its row; lanes 0-31 hold k=8s..8s+3 and lanes
// 32-63 hold k=8s+4..8s+7, so the j-th register of both operands forms the k-pair {8s+j, 8s+4+j} of
// one 32x32x2 MFMA - the sum over k is order-free, so no shuffling is needed.
#include "ttk_common.h"
#include "conv_geom.h"
#include <type_traits>
#include <stdlib.h>
#include <string.h>

namespace ttk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = TTK_GEMM_BLOCK_M;  // 128 rows of M per workgroup
constexpr int BKT = 32;               // contraction slice per LDS stage
static_assert(BKT == kCB, "one LDS stage = one channel block of the A operand");
constexpr int LDP = BKT + 4;          // padded LDS row (floats)

enum { MODE_FWD = 0, MODE_DGRAD = 1 };

// STAGES = 1: the whole contraction is one LDS stage (K = 32, the first pointwise layer): half the LDS, and with the
// 168-VGPR cap three workgroups per CU instead of two for this HBM-bound layer.
// T: storage of activations (y); TO: storage of this kernel's A0 operand and output - activations in the forward pass,
// activation gradients in the data gradient
template <int BN, int WM, int WN, int MODE, int STAGES, typename T, typename TO>
__global__ void __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(STAGES == 1 ? 4 : 1, STAGES == 1 ? 4 : 8)))
pw_gemm_k(const TO* __restrict__ A0, const T* __restrict__ A1,
                                                     const float* __restrict__ bnA, const float* __restrict__ Bm,
                                                     TO* __restrict__ out, const T* __restrict__ E0,
                                                     const float* __restrict__ bnE, float* __restrict__ part, int64_t M,
                                                     int K, int Nout) {
  // bnA: BatchNorm block [TTK_BN_ROWS][K] of the layer that produced the A operand (contraction channels);
  // bnE (data-gradient mode): block [TTK_BN_ROWS][Nout] of the layer whose ReLU masks the output.
  constexpr int NT = WM * WN * 64;  // threads: 4 or 8 waves
  static_assert(NT == 256 || NT == 512, "4 or 8 waves");
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile");
  constexpr int ROWS_PER_PASS = NT / 8;  // 8 lanes x 16 B cover one 32-float row of a stage
  constexpr int A_PASSES = BM / ROWS_PER_PASS;
  constexpr int B_PASSES = (BN + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
  // One LDS allocation: operand stages during the main loop, the C tile + reduction scratch afterwards.
  constexpr int LDC = BN + 4;
  constexpr int kStageFloats = STAGES * (BM + BN) * LDP;
  constexpr int kEpiFloats = BM * LDC + (NT / 64) * 2 * BN;
  constexpr int kSmemFloats = kStageFloats > kEpiFloats ? kStageFloats : kEpiFloats;
  __shared__ __attribute__((aligned(16))) float smem[kSmemFloats];
  float (*As)[BM][LDP] = reinterpret_cast<float (*)[BM][LDP]>(smem);
  float (*Bs)[BN][LDP] = reinterpret_cast<float (*)[BN][LDP]>(smem + STAGES * BM * LDP);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order: workgroup ids are dealt round-robin over the 8 XCDs (each with its own L2), so
  // ids b and b+8 share an L2.  Give every XCD a CONTIGUOUS range of tiles (column tiles of one M tile are
  // adjacent), so the A rows of an M tile are fetched into one L2 once instead of once per column tile.
  // Bijective for any grid size; placement only affects speed.
  const unsigned G = gridDim.x, Lid = blockIdx.x, NB = Nout / BN;
  const unsigned xq = G / 8, xr = G % 8, xcd = Lid % 8;
  const unsigned tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + Lid / 8;
  const unsigned bx = tile % NB, by = tile / NB;
  const int64_t m0 = (int64_t)by * BM;
  const int n0 = bx * BN;
  const int lrow = tid >> 3, kq = (tid & 7) * 4;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra0[A_PASSES], ra1[MODE == MODE_DGRAD ? A_PASSES : 1], rb[B_PASSES];
  const int nk = K / BKT;

  auto load_tile = [&](int kt) {
    const int k0 = kt * BKT + kq;
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      const int64_t row = m0 + p * ROWS_PER_PASS + lrow;
      const size_t ao = ((size_t)kt * M + row) * kCB + kq;  // channel block kt of the A operand (BKT == kCB), row `row`
      ra0[p] = (row < M) ? Act<TO>::ldnt(A0 + ao) : f4(0.f);
      if constexpr (MODE == MODE_DGRAD) ra1[p] = (row < M) ? Act<T>::ldnt(A1 + ao) : f4(0.f);
    }
#pragma unroll
    for (int p = 0; p < B_PASSES; ++p) {
      const int row = p * ROWS_PER_PASS + lrow;
      if (row < BN) rb[p] = ld4(Bm + (size_t)(n0 + row) * K + k0);
    }
  };
  auto store_tile = [&](int kt, int buf) {
    const int k0 = kt * BKT + kq;
    float4 q0, q1, q2, q3 = f4(0.f);
    if constexpr (MODE == MODE_FWD) {
      q0 = ld4(bnA + TTK_BN_SCALE * K + k0); q1 = ld4(bnA + TTK_BN_MEAN * K + k0); q2 = ld4(bnA + TTK_BN_BETA * K + k0);
    } else {
      q0 = ld4(bnA + TTK_BN_GA * K + k0); q1 = ld4(bnA + TTK_BN_GMEAN * K + k0);
      q2 = ld4(bnA + TTK_BN_GB * K + k0); q3 = ld4(bnA + TTK_BN_MEAN * K + k0);
    }
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      const int64_t row = m0 + p * ROWS_PER_PASS + lrow;
      float4 v;
      if constexpr (MODE == MODE_FWD) v = relu4(fma4(q0, sub4(ra0[p], q1), q2));
      else v = fma4(q0, sub4(ra0[p], q1), mul4(q2, sub4(ra1[p], q3)));
      if (row >= M) v = f4(0.f);
      st4(&As[buf][p * ROWS_PER_PASS + lrow][kq], v);
    }
#pragma unroll
    for (int p = 0; p < B_PASSES; ++p) {
      const int row = p * ROWS_PER_PASS + lrow;
      if (row < BN) st4(&Bs[buf][row][kq], rb[p]);
    }
  };

  load_tile(0);
  store_tile(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & (STAGES - 1);
    if (kt + 1 < nk) load_tile(kt + 1);
    const int fr = lane & 31, fk = 4 * (lane >> 5);
#pragma unroll
    for (int ks = 0; ks < BKT / 8; ++ks) {
      // the next tile's transform + ds_write go BETWEEN MFMA groups (after 3/4 of this tile's matrix work, so
      // its global loads have had ~3000 cycles to land): their VALU/LDS issue hides under the running MFMAs
      // instead of forming a matrix-idle phase in front of the barrier.
      if (STAGES > 1 && ks == BKT / 8 - 1 && kt + 1 < nk) store_tile(kt + 1, buf ^ 1);
      float4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = ld4(&As[buf][wm * (BM / WM) + i * 32 + fr][8 * ks + fk]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = ld4(&Bs[buf][wn * (BN / WN) + j * 32 + fr][8 * ks + fk]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
  }

  // ---- epilogue through LDS.  C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  // The accumulators are parked in a [BM][BN+4] LDS image (the operand stages are dead; the loop ended
  // with a barrier) and re-read row-wise, so that the mask operand, the BatchNorm partial sums and the
  // output use 16-byte global accesses (32 lanes x 16 B = one 512-byte row segment) instead of 64 scalar
  // loads + 64 scalar stores per lane with their latency exposed.
  float* Cs = smem;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int col = wn * (BN / WN) + j * 32 + (lane & 31);
        Cs[row * LDC + col] = acc[i][j][r];
      }
  __syncthreads();
  constexpr int QN = BN / 4;           // column quads
  constexpr int RG = NT / QN;          // row groups swept per pass
  const int c4 = tid % QN, rg = tid / QN;
  const int col = n0 + 4 * c4;
  float4 esc = f4(0.f), emean = f4(0.f), ebeta = f4(0.f);
  if constexpr (MODE == MODE_DGRAD) {
    esc = ld4(bnE + TTK_BN_SCALE * Nout + col); emean = ld4(bnE + TTK_BN_MEAN * Nout + col); ebeta = ld4(bnE + TTK_BN_BETA * Nout + col);
  }
  if constexpr (MODE == MODE_FWD) {
    if (bnE) emean = ld4(bnE + col);  // forward: bnE is the statistics pivot [Nout] (ttk.h), the sums are those of y - pivot
  }
  float4 s1 = f4(0.f), s2 = f4(0.f);
#pragma unroll 4
  for (int row = rg; row < BM; row += RG) {
    const int64_t grow = m0 + row;
    if (grow >= M) break;
    float4 v = ld4(Cs + row * LDC + 4 * c4);
    const size_t o = act_off(grow, col, M);
    if constexpr (MODE == MODE_FWD) {
      v = Act<TO>::round(v);  // statistics of what is stored
      Act<TO>::st(out + o, v);
      v = sub4(v, emean);
      s1 = add4(s1, v);
      s2 = fma4(v, v, s2);
    } else {
      const float4 yc = sub4(Act<T>::ldnt(E0 + o), emean);
      v = Act<TO>::round(mask4(v, fma4(esc, yc, ebeta)));
      Act<TO>::st(out + o, v);
      s1 = add4(s1, v);
      s2 = fma4(v, yc, s2);
    }
  }
  if (part) {
    // fold the row groups that live in one wave (lanes QN apart), then the 4 waves in fixed order
    for (int off = QN; off < kWave; off <<= 1) {
      s1.x += __shfl_xor(s1.x, off); s1.y += __shfl_xor(s1.y, off); s1.z += __shfl_xor(s1.z, off); s1.w += __shfl_xor(s1.w, off);
      s2.x += __shfl_xor(s2.x, off); s2.y += __shfl_xor(s2.y, off); s2.z += __shfl_xor(s2.z, off); s2.w += __shfl_xor(s2.w, off);
    }
    float* red = smem + BM * LDC;  // [4 waves][2][BN]
    if (lane < QN) {  // QN <= 32
      st4(red + (wave * 2 + 0) * BN + 4 * c4, s1);
      st4(red + (wave * 2 + 1) * BN + 4 * c4, s2);
    }
    __syncthreads();
    float* prow = part + (size_t)by * 2 * Nout;
    for (int i = tid; i < 2 * BN; i += NT) {
      const int which = i / BN, c = i % BN;
      float a = 0.f;
      for (int w = 0; w < NT / kWave; ++w) a += red[(w * 2 + which) * BN + c];
      prow[(size_t)which * Nout + n0 + c] = a;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: contraction over M.  Both operand tiles stay in their NATURAL layout
// ([32 rows of M][channels], channels contiguous): the MFMA wants, for A, lane (i=l&31, k=l>>5) =
// dY[m=2s+k][n=i] and for B lane (k, j) = a[m=2s+k][c=j], i.e. 32 consecutive floats of one row per
// half-wave: conflict-free ds_read_b32 with no transpose anywhere.
// Work split: grid.x = output tiles of dW, grid.y = slices of M; each wave may additionally own a
// slice of the 32-row stage (WS) when the dW tile is too small to feed 4 waves.
// ---------------------------------------------------------------------------------------------
template <int BN, int BK, int WR, int WC, int WS, typename T, typename TG>
__global__ void __launch_bounds__(kBlock) pw_wgrad_k(const TG* __restrict__ G, const T* __restrict__ Y,
                                                      const float* __restrict__ bn_pw, const T* __restrict__ Ydw,
                                                      const float* __restrict__ bn_dw, float* __restrict__ dW,
                                                      float* __restrict__ partial, int64_t M, int Cin, int Cout,
                                                      int64_t rows_per_slice) {
  static_assert(WR * WC * WS == 4, "4 waves");
  constexpr int TR = BN / WR / 32, TC = BK / WC / 32;
  constexpr int MS = 32;  // rows of M per stage
  constexpr int D_PASSES = (MS * BN / 4 + kBlock - 1) / kBlock;
  constexpr int A_PASSES = (MS * BK / 4 + kBlock - 1) / kBlock;
  __shared__ __attribute__((aligned(16))) float Ds[2][MS][BN + 4];
  __shared__ __attribute__((aligned(16))) float As[2][MS][BK + 4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ws = wave % WS, wc = (wave / WS) % WC, wr = wave / (WS * WC);
  const int tiles_k = Cin / BK;
  const int n0 = (blockIdx.x / tiles_k) * BN, k0 = (blockIdx.x % tiles_k) * BK;
  const int64_t m_begin = (int64_t)blockIdx.y * rows_per_slice;
  const int64_t m_end = (m_begin + rows_per_slice < M) ? m_begin + rows_per_slice : M;
  if (m_begin >= m_end) return;

  f32x16 acc[TR][TC];
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 rg[D_PASSES], ry[D_PASSES], ra[A_PASSES];
  auto load_stage = [&](int64_t ms) {
#pragma unroll
    for (int p = 0; p < D_PASSES; ++p) {
      const int f = p * kBlock + tid;
      const int row = f / (BN / 4), q = f % (BN / 4);
      const int64_t m = ms + row;
      if (row < MS && m < m_end) {
        rg[p] = Act<TG>::ldnt(G + act_off(m, n0 + 4 * q, M));
        ry[p] = Act<T>::ldnt(Y + act_off(m, n0 + 4 * q, M));
      } else {
        rg[p] = f4(0.f);
        ry[p] = f4(0.f);
      }
    }
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      const int f = p * kBlock + tid;
      const int row = f / (BK / 4), q = f % (BK / 4);
      const int64_t m = ms + row;
      ra[p] = (row < MS && m < m_end) ? Act<T>::ldnt(Ydw + act_off(m, k0 + 4 * q, M)) : f4(0.f);
    }
  };
  auto store_stage = [&](int64_t ms, int buf) {
#pragma unroll
    for (int p = 0; p < D_PASSES; ++p) {
      const int f = p * kBlock + tid;
      const int row = f / (BN / 4), q = f % (BN / 4);
      if (row < MS) {
        const int c = n0 + 4 * q;
        float4 v = BnGrad4::load(bn_pw, Cout, c).dy(rg[p], ry[p]);
        if (ms + row >= m_end) v = f4(0.f);
        st4(&Ds[buf][row][4 * q], v);
      }
    }
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
      const int f = p * kBlock + tid;
      const int row = f / (BK / 4), q = f % (BK / 4);
      if (row < MS) {
        const int c = k0 + 4 * q;
        float4 v = BnApply4::load(bn_dw, Cin, c).act(ra[p]);
        if (ms + row >= m_end) v = f4(0.f);
        st4(&As[buf][row][4 * q], v);
      }
    }
  };

  load_stage(m_begin);
  store_stage(m_begin, 0);
  __syncthreads();
  int it = 0;
  for (int64_t ms = m_begin; ms < m_end; ms += MS, ++it) {
    const int buf = it & 1;
    const bool more = ms + MS < m_end;
    if (more) load_stage(ms + MS);
    constexpr int PAIRS = MS / 2 / WS;  // m-pairs this wave multiplies per stage
#pragma unroll
    for (int s = 0; s < PAIRS; ++s) {
      const int mrow = 2 * (ws * PAIRS + s) + (lane >> 5);
      float a_op[TR], b_op[TC];
#pragma unroll
      for (int i = 0; i < TR; ++i) a_op[i] = Ds[buf][mrow][wr * (BN / WR) + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < TC; ++j) b_op[j] = As[buf][mrow][wc * (BK / WC) + j * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < TR; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_op[i], b_op[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_stage(ms + MS, buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TR; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wr * (BN / WR) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int k = k0 + wc * (BK / WC) + j * 32 + (lane & 31);
        if (WS == 1 && partial) partial[(size_t)blockIdx.y * Cout * Cin + (size_t)n * Cin + k] = acc[i][j][r];  // deterministic mode
        else atomicAdd((partial ? partial + (size_t)blockIdx.y * Cout * Cin : dW) + (size_t)n * Cin + k, acc[i][j][r]);
      }
}

__global__ void __launch_bounds__(kBlock) transpose_k(const float* __restrict__ in, float* __restrict__ out, int rows,
                                                       int cols) {
  __shared__ float t[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = ty; i < 32; i += 8)
    if (r0 + i < rows && c0 + tx < cols) t[i][tx] = in[(size_t)(r0 + i) * cols + c0 + tx];
  __syncthreads();
  for (int i = ty; i < 32; i += 8)
    if (c0 + i < cols && r0 + tx < rows) out[(size_t)(c0 + i) * rows + r0 + tx] = t[tx][i];
}

static bool pw_shape_ok(int64_t M, int Cin, int Cout) {
  auto p2 = [](int v) { return v >= 32 && v <= 1024 && (v & (v - 1)) == 0; };
  return M > 0 && p2(Cin) && p2(Cout);
}

// pwconv_f16.hip: the compute-bound shapes on the fp16 pipe with 2-piece operand splits (three products) - the default
template <int MODE, typename T, typename TO>
bool launch_f16_gemm(const TO* A0, const T* A1, const float* bnA, const float* Bm, TO* out, const T* E0,
                     const float* bnE, float* part, int64_t M, int K, int Nout, void* planes, float* wmax, hipStream_t st);
bool f16_gemm_shape(int K, int Nout);
template <typename T, typename TG>
bool launch_f16_wgrad(const TG* g, const T* y, const float* bn_pw, const T* ydw, const float* bn_dw, float* dw,
                      float* partial, int64_t M, int Cin, int Cout, hipStream_t st);
size_t f16_wgrad_partial_bytes(int64_t M, int Cin, int Cout);

// pwconv_r.hip: the wide layers (K >= 128, Nout a multiple of 256) in row-block form: 8 consumer waves, variable tile height,
// LDS-DMA weight planes in the [K/16][Nout][16] layout (conv_geom.h: r_plane_index)
template <int MODE, typename T, typename TO>
bool launch_f16r_gemm(const TO* A0, const T* A1, const float* bnA, const float* Bm, TO* out, const T* E0, const float* bnE, float* part,
                      int64_t M, int K, int Nout, void* planes, float* wmax, hipStream_t st);

// pwconv_r.hip: weight gradient of the layers with Cin, Cout multiples of 256 on 256 x 256 tiles with transposed LDS fragment reads; always
// reduces through `partial` (scratch of f16t_wgrad_scratch_bytes) and a fixed-order fold
template <typename T, typename TG>
bool launch_f16t_wgrad(const TG* g, const T* y, const float* bn_pw, const T* ydw, const float* bn_dw, float* dw, float* partial, int64_t M,
                       int Cin, int Cout, hipStream_t st);
size_t f16t_wgrad_scratch_bytes(int64_t M, int Cin, int Cout);

// (round 1's bf16 x 3 split kernels were removed in round 3: gemm_mode() never returns GEMM_BF16X3, these never launch)
template <int MODE>
inline bool launch_split_gemm(const float*, const float*, const float*, const float*, float*, const float*, const float*, float*, int64_t, int, int, void*,
                              hipStream_t) { return false; }
inline bool split_gemm_shape(int, int) { return false; }
inline bool launch_split_wgrad(const float*, const float*, const float*, const float*, const float*, float*, int64_t, int, int, hipStream_t) { return false; }

// Layout of a prepared weight block of n = Cin*Cout elements.  fp16 / fp32 modes: [forward operand 4n][data-gradient
// operand 4n][header: |w| maximum] - an operand is two fp16 planes or, for the shapes that stay on the fp32 kernels,
// the fp32 rows; bf16 mode: [forward 6n][data gradient 6n].
// (prep_bwd_offset / prep_hdr_offset: conv_geom.h - shared with the fused backward kernels of pw_bwd_fused.hip)

template <int MODE, typename T, typename TO>
static bool launch_gemm(const TO* A0, const T* A1, const float* bnA, const float* Bm, TO* out, const T* E0,
                        const float* bnE, float* part, int64_t M, int K, int Nout, void* region, float* hdr, hipStream_t st) {
  const int mode = gemm_mode();
  if constexpr (std::is_same<T, float>::value && std::is_same<TO, float>::value) {
    if (mode == GEMM_F16X2 && launch_f16y_gemm<MODE == MODE_FWD ? 0 : 1>(A0, A1, bnA, Bm, out, E0, bnE, part, M, K, Nout, region, hdr, st)) return true;
    if (mode == GEMM_F16X2 && launch_f16x_gemm<MODE == MODE_FWD ? 0 : 1>(A0, A1, bnA, Bm, out, E0, bnE, part, M, K, Nout, region, hdr, st)) return true;
  }
  if (mode == GEMM_F16X2 && launch_f16r_gemm<MODE, T, TO>(A0, A1, bnA, Bm, out, E0, bnE, part, M, K, Nout, region, hdr, st)) return true;
  if (mode == GEMM_F16X2 && launch_f16_gemm<MODE, T, TO>(A0, A1, bnA, Bm, out, E0, bnE, part, M, K, Nout, region, hdr, st)) return true;
  if constexpr (!Act<T>::kBf16 && !Act<TO>::kBf16) {
    if (mode == GEMM_BF16X3 && launch_split_gemm<MODE>(A0, A1, bnA, Bm, out, E0, bnE, part, M, K, Nout, region, st)) return true;
  } else {
    if (mode == GEMM_BF16X3) return false;  // the 3-piece bf16 kernels read fp32 activations only
  }
  if (!Bm) Bm = static_cast<const float*>(region);  // prepared operand of a shape that stays on the fp32 kernels
  const dim3 blk(kBlock);
  const unsigned gm = (unsigned)ceil_div(M, BM);
  if (Nout >= 128)
    hipLaunchKernelGGL((pw_gemm_k<128, 2, 2, MODE, 2, T, TO>), dim3((Nout / 128) * gm), blk, 0, st, A0, A1, bnA, Bm, out, E0, bnE, part, M, K,
                       Nout);
  else if (Nout == 64 && K == BKT)
    hipLaunchKernelGGL((pw_gemm_k<64, 2, 2, MODE, 1, T, TO>), dim3(gm), blk, 0, st, A0, A1, bnA, Bm, out, E0, bnE, part, M, K, Nout);
  else if (Nout == 64)
    hipLaunchKernelGGL((pw_gemm_k<64, 2, 2, MODE, 2, T, TO>), dim3(gm), blk, 0, st, A0, A1, bnA, Bm, out, E0, bnE, part, M, K, Nout);
  else
    hipLaunchKernelGGL((pw_gemm_k<32, 4, 1, MODE, 2, T, TO>), dim3(gm), blk, 0, st, A0, A1, bnA, Bm, out, E0, bnE, part, M, K, Nout);
  return true;
}

// ---- all pointwise layers' weight operands (ttk_pwconv_prepare_weights) -----------------------------------------------
constexpr int kPrepMax = 16;
struct PrepArgs {
  const float* w[kPrepMax];
  unsigned char* out[kPrepMax];
  int cin[kPrepMax], cout[kPrepMax];
  int first_tile[kPrepMax + 1];  // 32x32 tiles of w, layer after layer
  int split_fwd[kPrepMax], split_bwd[kPrepMax];
  int n, mode;
};

// element (row, k) of a [rows][K] operand: fp32 in place, or its piece planes in the split kernels' [K/32][rows][32] order
// (bf16 mode: three exact pieces; fp16 mode: two round-to-nearest pieces of x * s)
__device__ __forceinline__ void prep_store(unsigned char* region, int split, int mode, float s, int row, int k, int rows, int K, float x) {
  const int64_t n = (int64_t)rows * K;
  if (!split) {
    reinterpret_cast<float*>(region)[(int64_t)row * K + k] = x;
    return;
  }
  uint16_t* q = reinterpret_cast<uint16_t*>(region);
  if (split == 3) {  // fragment-ordered image, the two planes of a block side by side
    const float xs = x * s;
    const _Float16 hh = (_Float16)xs;
    const _Float16 ll = (_Float16)(xs - (float)hh);
    q[x_plane_index(row, k, rows, 0)] = __builtin_bit_cast(uint16_t, hh);
    q[x_plane_index(row, k, rows, 1)] = __builtin_bit_cast(uint16_t, ll);
    return;
  }
  const int64_t idx = split == 2 ? r_plane_index(row, k, rows) : ((int64_t)(k >> 5) * rows + row) * 32 + (k & 31);
  if (mode == GEMM_BF16X3) {
    const float r1 = x - __uint_as_float(__float_as_uint(x) & 0xffff0000u);
    const float r2 = r1 - __uint_as_float(__float_as_uint(r1) & 0xffff0000u);
    q[idx] = (uint16_t)(__float_as_uint(x) >> 16);
    q[n + idx] = (uint16_t)(__float_as_uint(r1) >> 16);
    q[2 * n + idx] = (uint16_t)(__float_as_uint(r2) >> 16);
  } else {
    const float xs = x * s;
    const _Float16 hh = (_Float16)xs;
    const _Float16 ll = (_Float16)(xs - (float)hh);
    q[idx] = __builtin_bit_cast(uint16_t, hh);
    q[n + idx] = __builtin_bit_cast(uint16_t, ll);
  }
}

__device__ __forceinline__ int prep_layer(const PrepArgs& a) {
  int l = 0;
  while (l + 1 < a.n && (int)blockIdx.x >= a.first_tile[l + 1]) ++l;
  return l;
}

// |w| maximum of every layer, in two steps without atomics: kPrepParts workgroups per layer leave their maxima behind the
// block's header, and every workgroup of pw_prepare_weights_k folds the kPrepParts values of its layer.  (The first form - one
// workgroup per 32x32 tile probing and raising one slot per layer - spent 33 us on 12 400 agent-scope reads of 13 addresses.)
constexpr int kPrepParts = 64;
__device__ __forceinline__ float* prep_parts(const PrepArgs& a, int l) {
  return reinterpret_cast<float*>(a.out[l] + 8 * (int64_t)a.cin[l] * a.cout[l] + 64);
}
__global__ void __launch_bounds__(kBlock) pw_prepare_absmax_k(PrepArgs a) {
  __shared__ float sm[kBlock / kWave];
  const int l = blockIdx.y;
  const int64_t n4 = (int64_t)a.cin[l] * a.cout[l] / 4;
  const float4* w4 = reinterpret_cast<const float4*>(a.w[l]);
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)kPrepParts * kBlock) {
    const float4 v = w4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) prep_parts(a, l)[blockIdx.x] = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
}

__global__ void __launch_bounds__(kBlock) pw_prepare_weights_k(PrepArgs a) {
  __shared__ float t[32][33];
  const int l = prep_layer(a);
  const int tile = blockIdx.x - a.first_tile[l];
  const int Cin = a.cin[l], Cout = a.cout[l];
  const int tk = Cin / 32, r0 = (tile / tk) * 32, c0 = (tile % tk) * 32;  // r: output channel, c: input channel
  const int64_t n = (int64_t)Cin * Cout;
  const float* w = a.w[l];
  unsigned char* fwd = a.out[l];
  unsigned char* bwd = a.out[l] + (a.mode == GEMM_BF16X3 ? 6 : 4) * n;
  float s = 1.f;
  if (a.mode == GEMM_F16X2) {
    static_assert(kPrepParts == kWave, "one value per lane");
    float m = prep_parts(a, l)[threadIdx.x & 63];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if (tile == 0 && threadIdx.x == 0) *reinterpret_cast<float*>(a.out[l] + 8 * n) = m;  // the header the GEMMs take their scale from
    s = pow2_scale(m);
  }
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += kBlock / 32) {
    const float x = w[(int64_t)(r0 + i) * Cin + c0 + tx];
    t[i][tx] = x;
    prep_store(fwd, a.split_fwd[l], a.mode, s, r0 + i, c0 + tx, Cout, Cin, x);
  }
  __syncthreads();
  for (int i = ty; i < 32; i += kBlock / 32) prep_store(bwd, a.split_bwd[l], a.mode, s, c0 + i, r0 + tx, Cin, Cout, t[tx][i]);
}

}  // namespace ttk

using namespace ttk;

extern "C" {

int ttk_partial_rows_pwconv(int64_t M, int K, int Nout, int dgrad) {
  const int yrows = f16y_partial_rows(M, K, Nout, dgrad);
  if (yrows) return yrows;
  const int x = f16x_partial_rows(M, K, Nout, dgrad);
  if (x) return x;
  const int r = f16r_partial_rows(M, K, Nout, dgrad);
  return r ? r : (int)ceil_div(M, BM);
}
int ttk_pwconv_tile_rows(int64_t M, int K, int Nout, int dgrad) {
  const int x = f16x_tile_rows(M, K, Nout, dgrad);
  return x ? x : f16r_tile_rows(M, K, Nout, dgrad);
}

int ttk_pwconv1x1_fwd(const void* ydw, const float* bn_dw, const float* w, void* y, float* part, const float* pivot, int64_t M, int Cin,
                      int Cout, void* wsplit, int act_bf16, ttk_stream_t stream) {
  TTK_REQUIRE(ydw && bn_dw && y && (w || wsplit), "pwconv1x1_fwd: null pointer");
  TTK_REQUIRE(pw_shape_ok(M, Cin, Cout), "pwconv1x1_fwd: unsupported shape M=%lld Cin=%d Cout=%d (channels: powers of two in 32..1024)", (long long)M, Cin, Cout);
  TTK_REQUIRE(ceil_div(M, BM) <= 65535, "pwconv1x1_fwd: M=%lld too large for one launch", (long long)M);
  unsigned char* ws = static_cast<unsigned char*>(wsplit);
  const size_t n = (size_t)Cin * Cout;
  float* hdr = ws ? reinterpret_cast<float*>(ws + prep_hdr_offset(n)) : nullptr;
  bool ok = false;
  TTK_ACT_DISPATCH(act_bf16, ok = launch_gemm<MODE_FWD, ActT, ActT>((const ActT*)ydw, nullptr, bn_dw, w, (ActT*)y, nullptr, pivot, part, M, Cin, Cout, ws,
                                                              hdr, (hipStream_t)stream));
  TTK_REQUIRE(ok, "pwconv1x1_fwd: bf16 activations need TTK_GEMM=f16x2 (default) or f32mfma");
  TTK_LAUNCH_CHECK("pwconv1x1_fwd");
}

int ttk_pwconv1x1_bwd_data(const void* g, const void* y, const float* bn_pw, const float* wt, const void* ydw,
                           const float* bn_dw, void* g_dw, float* part, int64_t M, int Cin, int Cout, void* wsplit,
                           int act_bf16, ttk_stream_t stream) {
  TTK_REQUIRE(g && y && bn_pw && (wt || wsplit) && ydw && bn_dw && g_dw, "pwconv1x1_bwd_data: null pointer");
  TTK_REQUIRE(pw_shape_ok(M, Cin, Cout), "pwconv1x1_bwd_data: unsupported shape");
  TTK_REQUIRE(ceil_div(M, BM) <= 65535, "pwconv1x1_bwd_data: M too large for one launch");
  unsigned char* ws = static_cast<unsigned char*>(wsplit);
  const size_t n = (size_t)Cin * Cout;
  unsigned char* region = (ws && !wt) ? ws + prep_bwd_offset(n) : ws;  // data-gradient half of a prepared block
  float* hdr = ws ? reinterpret_cast<float*>(ws + prep_hdr_offset(n)) : nullptr;
  // contraction over Cout, output columns = Cin, B operand = wt[Cin][Cout]
  bool ok = false;
  TTK_ACT_DISPATCH(act_bf16, ok = launch_gemm<MODE_DGRAD, ActT, GradT>((const GradT*)g, (const ActT*)y, bn_pw, wt, (GradT*)g_dw, (const ActT*)ydw, bn_dw, part,
                                                                M, Cout, Cin, region, hdr, (hipStream_t)stream));
  TTK_REQUIRE(ok, "pwconv1x1_bwd_data: bf16 activations need TTK_GEMM=f16x2 (default) or f32mfma");
  TTK_LAUNCH_CHECK("pwconv1x1_bwd_data");
}

// slices of M of the fp32-MFMA weight-gradient kernels
static void fp32_wgrad_plan(int64_t M, int Cin, int Cout, int& bn, int& bk, int& tiles, int64_t& slices, int64_t& rows) {
  bn = Cout >= 128 ? 128 : Cout;
  bk = Cin >= 128 ? 128 : Cin;
  tiles = (Cout / bn) * (Cin / bk);
  slices = 1024 / tiles;
  const int64_t max_slices = ceil_div(M, 256);
  if (slices > max_slices) slices = max_slices;
  if (slices < 1) slices = 1;
  rows = ceil_div(ceil_div(M, slices), 32) * 32;
  slices = ceil_div(M, rows);
}

size_t ttk_pwconv_wgrad_scratch_bytes(int64_t M, int Cin, int Cout) {
  if (!pw_shape_ok(M, Cin, Cout) || gemm_mode() != GEMM_F16X2) return 0;
  return f16t_wgrad_scratch_bytes(M, Cin, Cout);
}

size_t ttk_pwconv_wgrad_partial_bytes(int64_t M, int Cin, int Cout) {
  if (!pw_shape_ok(M, Cin, Cout)) return 0;
  if (gemm_mode() == GEMM_F16X2) {
    const size_t t = f16t_wgrad_scratch_bytes(M, Cin, Cout);
    if (t) return t;
    const size_t b = f16_wgrad_partial_bytes(M, Cin, Cout);
    if (b) return b;
  } else if (gemm_mode() == GEMM_BF16X3 && Cin >= 128 && Cout >= 128 && (int64_t)Cin * Cout >= 128 * 256) {
    return 0;  // the bf16 kernels have no deterministic form
  }
  int bn, bk, tiles;
  int64_t slices, rows;
  fp32_wgrad_plan(M, Cin, Cout, bn, bk, tiles, slices, rows);
  return (size_t)slices * Cin * Cout * sizeof(float);
}

int ttk_pwconv1x1_bwd_weight(const void* g, const void* y, const float* bn_pw, const void* ydw, const float* bn_dw, float* dw,
                             float* partial, int64_t M, int Cin, int Cout, int act_bf16, ttk_stream_t stream) {
  TTK_REQUIRE(g && y && bn_pw && ydw && bn_dw && dw, "pwconv1x1_bwd_weight: null pointer");
  TTK_REQUIRE(pw_shape_ok(M, Cin, Cout), "pwconv1x1_bwd_weight: unsupported shape");
  TTK_REQUIRE(!(act_bf16 && gemm_mode() == GEMM_BF16X3), "pwconv1x1_bwd_weight: bf16 activations need TTK_GEMM=f16x2 (default) or f32mfma");
  if (gemm_mode() == GEMM_F16X2) {
    bool done = false;
    TTK_ACT_DISPATCH(act_bf16, done = launch_f16t_wgrad<ActT, GradT>((const GradT*)g, (const ActT*)y, bn_pw, (const ActT*)ydw, bn_dw, dw, partial, M, Cin, Cout,
                                                              (hipStream_t)stream));
    if (done) { TTK_LAUNCH_CHECK("pwconv1x1_bwd_weight"); }
    TTK_ACT_DISPATCH(act_bf16, done = launch_f16_wgrad<ActT, GradT>((const GradT*)g, (const ActT*)y, bn_pw, (const ActT*)ydw, bn_dw, dw, partial, M, Cin, Cout,
                                                             (hipStream_t)stream));
    if (done) { TTK_LAUNCH_CHECK("pwconv1x1_bwd_weight"); }
  }
  if (gemm_mode() == GEMM_BF16X3 && launch_split_wgrad((const float*)g, (const float*)y, bn_pw, (const float*)ydw, bn_dw, dw, M, Cin, Cout,
                                                      (hipStream_t)stream)) {
    TTK_LAUNCH_CHECK("pwconv1x1_bwd_weight");
  }
  int bn, bk, tiles;
  int64_t slices, rows;
  fp32_wgrad_plan(M, Cin, Cout, bn, bk, tiles, slices, rows);
  const dim3 grid(tiles, (unsigned)slices), blk(kBlock);
  hipStream_t st = (hipStream_t)stream;
  // deterministic mode: every slice of M owns partial[slice][Cout][Cin]; the two waves that share an output tile in the
  // WS = 2 variants add into a ZEROED slot (a + b = b + a: still reproducible); four-wave sharing (32 x 32) is not
  const bool shared = (bn == 64 && bk == 32) || (bn == 32 && bk == 64);
  if (partial && bn == 32 && bk == 32) partial = nullptr;
  if (partial && shared) (void)hipMemsetAsync(partial, 0, (size_t)slices * Cin * Cout * sizeof(float), st);
#define TTK_WG(BN_, BK_, WR_, WC_, WS_)                                                                                                 \
  hipLaunchKernelGGL((pw_wgrad_k<BN_, BK_, WR_, WC_, WS_, ActT, GradT>), grid, blk, 0, st, (const GradT*)g, (const ActT*)y, bn_pw, (const ActT*)ydw, bn_dw, \
                     dw, partial, M, Cin, Cout, rows)
  TTK_ACT_DISPATCH(act_bf16,
                   if (bn == 128 && bk == 128) TTK_WG(128, 128, 2, 2, 1);
                   else if (bn == 128 && bk == 64) TTK_WG(128, 64, 2, 2, 1);
                   else if (bn == 128 && bk == 32) TTK_WG(128, 32, 4, 1, 1);
                   else if (bn == 64 && bk == 128) TTK_WG(64, 128, 2, 2, 1);
                   else if (bn == 64 && bk == 64) TTK_WG(64, 64, 2, 2, 1);
                   else if (bn == 64 && bk == 32) TTK_WG(64, 32, 2, 1, 2);
                   else if (bn == 32 && bk == 128) TTK_WG(32, 128, 1, 4, 1);
                   else if (bn == 32 && bk == 64) TTK_WG(32, 64, 1, 2, 2);
                   else TTK_WG(32, 32, 1, 1, 4));
#undef TTK_WG
  if (partial) launch_fold_partials(partial, (int)slices, (int64_t)Cin * Cout, dw, 1, st);
  TTK_LAUNCH_CHECK("pwconv1x1_bwd_weight");
}

size_t ttk_pwconv_prepared_bytes(int Cin, int Cout) {
  const size_t n = (size_t)Cin * Cout;
  return prep_hdr_offset(n) + 64 + kPrepParts * sizeof(float);  // planes, header, workgroup maxima of the prepare step
}

int ttk_pwconv_prepare_weights(int n, const float* const* w, const int* cin, const int* cout, void* const* prepared,
                               ttk_stream_t stream) {
  TTK_REQUIRE(n > 0 && n <= kPrepMax && w && cin && cout && prepared, "pwconv_prepare_weights: bad arguments (1..16 layers)");
  PrepArgs a{};
  a.n = n;
  a.mode = gemm_mode();
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    TTK_REQUIRE(w[i] && prepared[i] && pw_shape_ok(1, cin[i], cout[i]), "pwconv_prepare_weights: layer %d: null pointer or unsupported shape", i);
    a.w[i] = w[i];
    a.out[i] = static_cast<unsigned char*>(prepared[i]);
    a.cin[i] = cin[i];
    a.cout[i] = cout[i];
    a.first_tile[i] = tiles;
    tiles += (cin[i] / 32) * (cout[i] / 32);
    // 0: fp32 rows; 1: piece planes [K/32][rows][32]; 2: the row-block kernels' planes [K/16][rows][16] (pwconv_r.hip); 3: the full-width kernels'
    // fragment-ordered image (pwconv_x.hip)
    a.split_fwd[i] = a.mode == GEMM_F16X2 ? ((f16y_gemm_shape(cin[i], cout[i], 0) || f16x_gemm_shape(cin[i], cout[i], 0)) ? 3 : f16r_gemm_shape(cin[i], cout[i], 0) ? 2 : f16_gemm_shape(cin[i], cout[i])) : (a.mode == GEMM_BF16X3 && split_gemm_shape(cin[i], cout[i]));
    a.split_bwd[i] = a.mode == GEMM_F16X2 ? ((f16y_gemm_shape(cout[i], cin[i], 1) || f16x_gemm_shape(cout[i], cin[i], 1)) ? 3 : f16r_gemm_shape(cout[i], cin[i], 1) ? 2 : f16_gemm_shape(cout[i], cin[i])) : (a.mode == GEMM_BF16X3 && split_gemm_shape(cout[i], cin[i]));
  }
  a.first_tile[n] = tiles;
  hipStream_t st = (hipStream_t)stream;
  if (a.mode == GEMM_F16X2) hipLaunchKernelGGL(pw_prepare_absmax_k, dim3(kPrepParts, n), dim3(kBlock), 0, st, a);
  hipLaunchKernelGGL(pw_prepare_weights_k, dim3(tiles), dim3(kBlock), 0, st, a);
  TTK_LAUNCH_CHECK("pwconv_prepare_weights");
}

int ttk_transpose(const float* in, float* out, int rows, int cols, ttk_stream_t stream) {
  TTK_REQUIRE(in && out && rows > 0 && cols > 0, "transpose: bad arguments");
  hipLaunchKernelGGL(transpose_k, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(kBlock), 0, (hipStream_t)stream, in, out, rows,
                     cols);
  TTK_LAUNCH_CHECK("transpose");
}

}  // extern "C"
