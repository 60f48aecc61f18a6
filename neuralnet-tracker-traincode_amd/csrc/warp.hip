// On-GPU affine-warp augmentation.  In the reference this runs per sample on
// the CPU inside DataLoader workers with OpenCV (datatransformation/batch/geometric.py:193-231); its pure
// torch formulation (tensors/image_geometric_torch.py:60-98: affine_grid + grid_sample, bilinear, zero
// padding, align_corners=False) is the semantic oracle for the image, and tensors/affinetrafo.py:37-148 for
// the labels.
//
//   view_roi   : GeneralFocusRoi._compute_view_roi (:108-157) + torch.round(...).to(int32) (:205)
//                INTEGER result, bit-exact: every float op is an explicitly rounded __f*_rn (no fma
//                contraction) in the reference's order; rintf = round-half-to-even like torch.round.
//   roi -> tr  : center_rotation(angle) @ range_remap(view_roi -> [0,N]^2)            (:159-177)
//   warp       : out[b,0,i,j] = bilinear(src_b, tr^-1 (j+.5, i+.5) - .5) * mul + add   (gather-bound)
//   labels     : coord / pose / roi / pt3d_68 under tr, then under the [0,N] -> [-1,1] normalisation
#include "head_math.h"
#include "ttk_common.h"

namespace ttk {

__global__ void view_roi_k(const float* __restrict__ roi, const float* __restrict__ f, const float* __restrict__ t, float bbs,
                           int B, int* __restrict__ out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float x0 = roi[4 * b], y0 = roi[4 * b + 1], x1 = roi[4 * b + 2], y1 = roi[4 * b + 3];
  const float rx = t[2 * b], ry = t[2 * b + 1];
  const float bw = __fsub_rn(x1, x0), bh = __fsub_rn(y1, y0);
  const float cx = __fmul_rn(0.5f, __fadd_rn(x1, x0)), cy = __fmul_rn(0.5f, __fadd_rn(y1, y0));
  const float size = __fmul_rn(fmaxf(bw, bh), f[b]);
  const float wx = __fadd_rn(__fmul_rn(0.5f, fabsf(__fsub_rn(size, bw))), __fmul_rn(bbs, fminf(size, bw)));
  const float wy = __fadd_rn(__fmul_rn(0.5f, fabsf(__fsub_rn(size, bh))), __fmul_rn(bbs, fminf(size, bh)));
  const float tx = __fmul_rn(wx, rx), ty = __fmul_rn(wy, ry);
  const float hs = __fmul_rn(size, 0.5f);
  out[4 * b + 0] = (int)rintf(__fadd_rn(__fsub_rn(cx, hs), tx));
  out[4 * b + 1] = (int)rintf(__fadd_rn(__fsub_rn(cy, hs), ty));
  out[4 * b + 2] = (int)rintf(__fadd_rn(__fadd_rn(cx, hs), tx));
  out[4 * b + 3] = (int)rintf(__fadd_rn(__fadd_rn(cy, hs), ty));
}

// tr = unnorm(N) @ rot(angle) @ norm(N) @ remap(view_roi -> [0,N]^2), row-major 2x3
__global__ void roi_transform_k(const int* __restrict__ vr, const float* __restrict__ angles, int B, int N,
                                float* __restrict__ tr) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float x0 = (float)vr[4 * b], y0 = (float)vr[4 * b + 1], x1 = (float)vr[4 * b + 2], y1 = (float)vr[4 * b + 3];
  const float sx = (float)N / (x1 - x0), sy = (float)N / (y1 - y0);
  const float ox = -x0 * sx, oy = -y0 * sy;  // remap
  const float a = angles ? angles[b] : 0.f, c = cosf(a), s = sinf(a), h = 0.5f * (float)N;
  // p -> (p - h)/h -> R -> *h + h   ==  R p + (h - R h)
  const float r00 = c, r01 = -s, r10 = s, r11 = c;
  const float t0 = h - (r00 * h + r01 * h), t1 = h - (r10 * h + r11 * h);
  float* m = tr + 6 * b;
  m[0] = r00 * sx; m[1] = r01 * sy; m[2] = r00 * ox + r01 * oy + t0;
  m[3] = r10 * sx; m[4] = r11 * sy; m[5] = r10 * ox + r11 * oy + t1;
}

template <typename T>
__device__ __forceinline__ float fetch(const T* img, int H, int W, int y, int x) {
  return (x >= 0 && x < W && y >= 0 && y < H) ? (float)img[(size_t)y * W + x] : 0.f;
}

// one thread per output pixel; consecutive lanes = consecutive output columns (coalesced store, gather load)
template <typename T>
__global__ void __launch_bounds__(kBlock) affine_warp_k(const T* __restrict__ src, int B, int Hs, int Ws,
                                                         const float* __restrict__ tr, float* __restrict__ out, int N, float mul,
                                                         float add) {
  const int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (idx >= (int64_t)B * N * N) return;
  const int j = (int)(idx % N), i = (int)((idx / N) % N), b = (int)(idx / ((int64_t)N * N));
  const float* m = tr + 6 * b;
  const float det = m[0] * m[4] - m[1] * m[3];
  const float inv = 1.f / det;
  const float px = (float)j + 0.5f - m[2], py = (float)i + 0.5f - m[5];
  const float u = (m[4] * px - m[1] * py) * inv - 0.5f;
  const float v = (-m[3] * px + m[0] * py) * inv - 0.5f;
  const float fu = floorf(u), fv = floorf(v);
  const int x0 = (int)fu, y0 = (int)fv;
  const float ax = u - fu, ay = v - fv;
  const T* img = src + (size_t)b * Hs * Ws;
  const float v00 = fetch(img, Hs, Ws, y0, x0), v01 = fetch(img, Hs, Ws, y0, x0 + 1);
  const float v10 = fetch(img, Hs, Ws, y0 + 1, x0), v11 = fetch(img, Hs, Ws, y0 + 1, x0 + 1);
  const float val = (v00 * (1.f - ax) + v01 * ax) * (1.f - ay) + (v10 * (1.f - ax) + v11 * ax) * ay;
  out[idx] = fmaf(val, mul, add);
}

// 68-landmark left/right partner under a horizontal mirror (facemodel/keypoints68.py:7-77)
__constant__ unsigned char kFlipMap[68] = {16, 15, 14, 13, 12, 11, 10, 9,  8,  7,  6,  5,  4,  3,  2,  1,  0,  26, 25, 24, 23, 22, 21,
                                           20, 19, 18, 17, 27, 28, 29, 30, 35, 34, 33, 32, 31, 45, 44, 43, 42, 47, 46, 39, 38, 37, 36,
                                           41, 40, 54, 53, 52, 51, 50, 49, 48, 59, 58, 57, 56, 55, 64, 63, 62, 61, 60, 67, 66, 65};

struct Aff {
  float a, b, tx, c, d, ty;
  __device__ __forceinline__ float det() const { return a * d - b * c; }
  __device__ __forceinline__ float scale() const { return sqrtf(a * a + b * b + c * c + d * d) * 0.70710678118654752440f; }
};

// labels of one sample under `m` (tensors/affinetrafo.py: transform_coord :107-114, transform_rot :117-148,
// transform_roi :91-104, transform_points/keypoints :37-88).  pts_in/pts_out may alias only if det >= 0.
__device__ void labels_under(const Aff m, float* coord, float* pose, float* roi, const float* pts_in, float* pts_out,
                             int lane) {
  const float det = m.det();
  if (lane == 0) {
    if (coord) {
      const float x = coord[0], y = coord[1];
      coord[0] = m.a * x + m.b * y + m.tx;
      coord[1] = m.c * x + m.d * y + m.ty;
      coord[2] = m.scale() * coord[2];
    }
    if (pose) {
      const float sg = det > 0.f ? 1.f : (det < 0.f ? -1.f : 0.f);
      const float alpha = atan2f(-m.b, m.d);
      const hm::Q z{0.f, 0.f, sinf(0.5f * alpha) * sg, cosf(0.5f * alpha)};
      hm::Q o = hm::qmul(z, hm::Q{pose[0], pose[1], pose[2], pose[3]});
      pose[0] = o.i; pose[1] = sg * o.j; pose[2] = sg * o.k; pose[3] = o.w;
    }
    if (roi) {
      const float xs[2] = {roi[0], roi[2]}, ys[2] = {roi[1], roi[3]};
      float lo0 = 3.4e38f, lo1 = 3.4e38f, hi0 = -3.4e38f, hi1 = -3.4e38f;
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) {
          const float px = m.a * xs[i] + m.b * ys[j] + m.tx, py = m.c * xs[i] + m.d * ys[j] + m.ty;
          lo0 = fminf(lo0, px); hi0 = fmaxf(hi0, px); lo1 = fminf(lo1, py); hi1 = fmaxf(hi1, py);
        }
      roi[0] = lo0; roi[1] = lo1; roi[2] = hi0; roi[3] = hi1;
    }
  }
  if (pts_in) {
    const float zs = sqrtf(fabsf(det));
    for (int p = lane; p < 68; p += 64) {
      const int q = det < 0.f ? kFlipMap[p] : p;  // out[p] = transformed in[flip_map[p]]
      const float x = pts_in[3 * q], y = pts_in[3 * q + 1], z = pts_in[3 * q + 2];
      pts_out[3 * p] = m.a * x + m.b * y + m.tx;
      pts_out[3 * p + 1] = m.c * x + m.d * y + m.ty;
      pts_out[3 * p + 2] = zs * z;
    }
  }
}

// one wave per sample: labels under tr[b], then (N > 0) under the pixel -> [-1,1] normalisation
__global__ void __launch_bounds__(kWave) affine_labels_k(const float* __restrict__ tr, int B, int N, float* coord, float* pose,
                                                          float* roi, const float* pts_in, float* pts_out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (b >= B) return;
  const float* t = tr + 6 * b;
  float* c = coord ? coord + 3 * b : nullptr;
  float* q = pose ? pose + 4 * b : nullptr;
  float* r = roi ? roi + 4 * b : nullptr;
  const float* pi = pts_in ? pts_in + (size_t)b * 204 : nullptr;
  float* po = pts_out ? pts_out + (size_t)b * 204 : nullptr;
  labels_under(Aff{t[0], t[1], t[2], t[3], t[4], t[5]}, c, q, r, pi, po, lane);
  if (N > 0) {
    __syncthreads();
    const float s = 2.f / (float)N;
    labels_under(Aff{s, 0.f, -1.f, 0.f, s, -1.f}, c, q, r, po, po, lane);
  }
}

}  // namespace ttk

using namespace ttk;

extern "C" {

int ttk_view_roi(const float* face_roi, const float* scales, const float* translations, float beyond_border_shift, int B,
                 int* view_roi, ttk_stream_t stream) {
  TTK_REQUIRE(face_roi && scales && translations && view_roi && B > 0, "view_roi: bad arguments");
  hipLaunchKernelGGL(view_roi_k, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, face_roi, scales, translations,
                     beyond_border_shift, B, view_roi);
  TTK_LAUNCH_CHECK("view_roi");
}

int ttk_roi_transform(const int* view_roi, const float* angles, int B, int N, float* tr, ttk_stream_t stream) {
  TTK_REQUIRE(view_roi && tr && B > 0 && N > 0, "roi_transform: bad arguments");
  hipLaunchKernelGGL(roi_transform_k, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, view_roi, angles, B, N, tr);
  TTK_LAUNCH_CHECK("roi_transform");
}

int ttk_affine_warp(const void* src, int src_is_u8, int B, int Hs, int Ws, const float* tr, float* out, int N, float mul,
                    float add, ttk_stream_t stream) {
  TTK_REQUIRE(src && tr && out && B > 0 && Hs > 0 && Ws > 0 && N > 0, "affine_warp: bad arguments");
  const unsigned grid = (unsigned)ceil_div((int64_t)B * N * N, kBlock);
  if (src_is_u8)
    hipLaunchKernelGGL(affine_warp_k<unsigned char>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                       (const unsigned char*)src, B, Hs, Ws, tr, out, N, mul, add);
  else
    hipLaunchKernelGGL(affine_warp_k<float>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, (const float*)src, B, Hs, Ws, tr,
                       out, N, mul, add);
  TTK_LAUNCH_CHECK("affine_warp");
}

int ttk_affine_labels(const float* tr, int B, int N, float* coord, float* pose, float* roi, const float* pts_in, float* pts_out,
                      ttk_stream_t stream) {
  TTK_REQUIRE(tr && B > 0, "affine_labels: bad arguments");
  TTK_REQUIRE((pts_in == nullptr) == (pts_out == nullptr) && pts_in != pts_out || !pts_in, "affine_labels: pts_in/pts_out must be two distinct buffers");
  hipLaunchKernelGGL(affine_labels_k, dim3(B), dim3(kWave), 0, (hipStream_t)stream, tr, B, N, coord, pose, roi, pts_in, pts_out);
  TTK_LAUNCH_CHECK("affine_labels");
}

}  // extern "C"
