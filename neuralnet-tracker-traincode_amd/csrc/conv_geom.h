// Shared between pwconv_split.hip and conv.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace ttk {

// A-operand forms (what the producers compute from the loaded rows) and epilogue forms
// AMODE_PLANES: the A operand arrives already split - A0 / A1 = the h / l fp16 planes [rows][Kc] of x * pow2_scale(bound)
// (ttk_bn_bwd_apply writes dy that way) - and the producers only move it (fp16 kernels)
enum { AMODE_BNRELU = 0, AMODE_BNGRAD = 1, AMODE_PLAIN = 2, AMODE_PLANES = 3 };
enum { EMODE_STATS = 0, EMODE_MASK = 1, EMODE_PLAIN = 2 };

// Implicit-GEMM convolution (ResNet 3x3 / strided 1x1, backbones/resnet.py): the GEMM rows enumerate the pixels of a
// grid (Hg x Wg per image) and the contraction runs over (tap, channel): for tap (kh, kw) row (n, gh, gw) reads the
// source pixel   forward:    (gh*stride - pad + kh, gw*stride - pad + kw)            [grid = output, source = input]
//                transposed: ((gh + pad - kh)/stride, (gw + pad - kw)/stride) if divisible  [grid = input, source = output]
// of a [n][Hs][Ws][Kc] tensor, or contributes zero outside it.  The B operand is [tap][Nout][Kc].
// Parity classes (par = 1; fp16 kernels, transposed stride-2 only): a grid pixel (gh, gw) of a stride-2 data gradient is
// reached by the taps with kh = gh + pad (mod 2) only - 1, 2, 2 or 4 of the 9 taps of a 3x3 kernel, one of the four
// classes of a strided 1x1 kernel - so the launch enumerates the pixels class by class ((gh & 1, gw & 1) = (1,1), (1,0),
// (0,1), (0,0): longest contraction first) and every tile contracts over its class's taps only.  ctile[c] = first tile
// of class c, nimg = images (the rows of class c are nimg x its pixels).
struct ConvGeom {
  int Hs, Ws, Hg, Wg, stride, pad, KW, Kc, transposed;
  int par, nimg, ctile[4];
};

// TTK_GEMM=f32mfma keeps every pointwise conv on v_mfma_f32_32x32x2_f32, bf16x3 selects the 3-piece bf16 split (A/B
// timing and numerics comparisons; the ResNet18 convolutions have no fp32 form and take bf16x3 for both)
enum { GEMM_F16X2 = 0, GEMM_BF16X3 = 1, GEMM_F32 = 2 };
inline int gemm_mode() {
  static const int mode = [] {
    const char* e = exp_env("TTK_GEMM");
    if (e && strcmp(e, "f32mfma") == 0) return (int)GEMM_F32;
    if (e && strcmp(e, "bf16x3") == 0)
      fprintf(stderr, "libttk_hip: TTK_GEMM=bf16x3 (round 1's six-product bf16 split, csrc/pwconv_split.hip) was removed in round 3 - it is in the "
                      "git history before the channel-block activation layout; using the fp16 kernels\n");
    return (int)GEMM_F16X2;
  }();
  return mode;
}

// Round 1's six-product bf16 split (csrc/pwconv_split.hip, TTK_GEMM=bf16x3) was removed in round 3; gemm_mode() never returns
// GEMM_BF16X3 and these never launch.
inline bool launch_conv_gemm(int, int, const float*, const float*, const float*, const uint16_t*, float*, const float*, const float*, float*, int64_t,
                             int, int, const ConvGeom&, hipStream_t) { return false; }
inline bool launch_conv_wgrad(const float*, const float*, const float*, const float*, float*, int64_t, int, int, const ConvGeom&, hipStream_t) { return false; }

// Layout of a prepared weight block of n = Cin * Cout elements (ttk_pwconv_prepare_weights): [forward operand][data-gradient operand]
// [header: |w| maximum ...].  Every reader of the block takes the offsets from here.
inline size_t prep_bwd_offset(size_t n) { return 4 * n; }
inline size_t prep_hdr_offset(size_t n) { return 8 * n; }

// fp16-pipe forms (pwconv_f16.hip): Bq = two fp16 planes scaled by pow2_scale(*wmax); a_bound = bound of a plain A operand
bool launch_conv_gemm16(int amode, int emode, const float* A0, const float* A1, const float* bnA, const float* a_bound, const uint16_t* Bq,
                        const float* wmax, float* out, const float* E0, float* bnE, float* part, int64_t M, int K, int Nout,
                        const ConvGeom& geo, hipStream_t st);
bool launch_conv_wgrad16(const float* g, const float* y, const float* bn, const float* a_in, const float* a_bound, float* dw, float* partial,
                         int64_t M, int Cout, int taps, const ConvGeom& geo, hipStream_t st);
size_t conv_wgrad16_partial_bytes(int64_t M, int Cout, int ncols, int taps);

// Row-block GEMMs (pwconv_r.hip): element (row, k) of a [rows][K] weight operand inside one piece plane [K/16][rows][16] whose two
// 16-byte chunks per row are swapped where (row >> 3) & 1 - the LDS image of a k16 stage, so that LDS-DMA copies it linearly
__host__ __device__ inline int64_t r_plane_index(int row, int k, int rows) {
  return ((int64_t)(k >> 4) * rows + row) * 16 + ((((k >> 3) & 1) ^ ((row >> 3) & 1)) << 3) + (k & 7);
}
// element (row n, k) of a [rows][K] weight operand inside the fragment-ordered image of ttk_pwconv_prepare_weights (split code 3):
// [k32 step][16-row block][plane][lane = 16 (k chunk) + row][8 k] fp16 - a wave's fragment of one block and plane is 1 KB, lane-linear
__host__ __device__ inline int64_t x_plane_index(int row, int k, int rows, int plane) {
  const int ks = k >> 5, q = (k >> 3) & 3, cb = row >> 4, r = row & 15;
  return (((int64_t)ks * (rows >> 4) + cb) * 2 + plane) * 512 + (q * 16 + r) * 8 + (k & 7);
}

// Full-width GEMMs (pwconv_x.hip, round 6): they take the wide shapes in the product; the row-block kernels stay for experiment builds (TTK_GEMM_X=0)
bool f16x_gemm_shape(int K, int Nout, int dgrad);
int f16x_partial_rows(int64_t M, int K, int Nout, int dgrad);
int f16x_tile_rows(int64_t M, int K, int Nout, int dgrad);
template <int MODE>
bool launch_f16x_gemm(const float* A0, const float* A1, const float* bnA, const float* Bm, float* out, const float* E0, const float* bnE, float* part, int64_t M,
                      int K, int Nout, void* planes, float* wmax, hipStream_t st);
// Streaming GEMMs of the narrow HBM-bound layers (pwconv_y.hip, round 6)
bool f16y_gemm_shape(int K, int Nout, int dgrad);
int f16y_partial_rows(int64_t M, int K, int Nout, int dgrad);
template <int MODE>
bool launch_f16y_gemm(const float* A0, const float* A1, const float* bnA, const float* Bm, float* out, const float* E0, const float* bnE, float* part, int64_t M,
                      int K, int Nout, void* planes, float* wmax, hipStream_t st);
bool f16r_gemm_shape(int K, int Nout, int dgrad);
int f16r_partial_rows(int64_t M, int K, int Nout, int dgrad);
int f16r_tile_rows(int64_t M, int K, int Nout, int dgrad);

}  // namespace ttk
