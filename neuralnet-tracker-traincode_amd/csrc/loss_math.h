// Per-sample loss values and their hand-derived gradients (host + device header, see head_math.h).
//
// Reference (paths relative to trackertraincode/neuralnets/):
//   losses.py:42-50,67-97,116-173          QuatPoseLoss, PoseXY/Size/Box/ShapeParameter L2, Points3dLoss,
//                                          QuaternionNormalizationSoftConstraint
//   losses.py:100-113 + modelcomponents.py:278-290   ShapePlausibilityLoss (10-component diagonal GMM, float64)
//   negloglikelihood.py:100-126,129-177,245-274     uniform-mixed MVN / Normal negative log-likelihoods
//   torchquaternion.py:187-218             to_rotvec / rotation_delta / positivereal
#pragma once
#include "head_math.h"

namespace ttk {
namespace lm {

using hm::Q;

constexpr float kHalfLog2Pi = 0.91893853320467274178f;

// ---- 1 - (q.t)^2  (torchquaternion.distance) -------------------------------------------------------
TTK_HD float rot_loss(const float q[4], const float t[4]) {
  const float d = q[0] * t[0] + q[1] * t[1] + q[2] * t[2] + q[3] * t[3];
  return 1.f - d * d;
}
TTK_HD void rot_loss_bwd(const float q[4], const float t[4], float gv, float gq[4]) {
  const float d = q[0] * t[0] + q[1] * t[1] + q[2] * t[2] + q[3] * t[3];
  const float f = -2.f * d * gv;
  gq[0] = f * t[0]; gq[1] = f * t[1]; gq[2] = f * t[2]; gq[3] = f * t[3];
}

// ---- (1 - |qu|)^2 -----------------------------------------------------------------------------------
TTK_HD float quatreg_loss(const float q[4]) {
  const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  return (1.f - n) * (1.f - n);
}
TTK_HD void quatreg_loss_bwd(const float q[4], float gv, float gq[4]) {
  const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  const float f = n > 0.f ? -2.f * (1.f - n) / n * gv : 0.f;
  gq[0] = f * q[0]; gq[1] = f * q[1]; gq[2] = f * q[2]; gq[3] = f * q[3];
}

// ---- log N(delta; 0, L L^T), L lower-triangular 3x3 row-major --------------------------------------
struct Mvn3 {
  float y[3], lp;
};
TTK_HD Mvn3 mvn_logprob(const float d[3], const float L[9]) {
  Mvn3 r;
  r.y[0] = d[0] / L[0];
  r.y[1] = (d[1] - L[3] * r.y[0]) / L[4];
  r.y[2] = (d[2] - L[6] * r.y[0] - L[7] * r.y[1]) / L[8];
  const float maha = r.y[0] * r.y[0] + r.y[1] * r.y[1] + r.y[2] * r.y[2];
  r.lp = -0.5f * (6.f * kHalfLog2Pi + maha) - (logf(L[0]) + logf(L[4]) + logf(L[8]));
  return r;
}
// glp = dLoss/dlp.  Writes gd[3], gL[9] (upper triangle = 0).
TTK_HD void mvn_logprob_bwd(const float L[9], const Mvn3& f, float glp, float gd[3], float gL[9]) {
  float gy0 = -f.y[0] * glp, gy1 = -f.y[1] * glp, gy2 = -f.y[2] * glp;
  const float i22 = 1.f / L[8], i11 = 1.f / L[4], i00 = 1.f / L[0];
  gd[2] = gy2 * i22;
  gL[6] = -gy2 * f.y[0] * i22;
  gL[7] = -gy2 * f.y[1] * i22;
  gL[8] = -gy2 * f.y[2] * i22 - glp * i22;
  gy0 -= gy2 * L[6] * i22;
  gy1 -= gy2 * L[7] * i22;
  gd[1] = gy1 * i11;
  gL[3] = -gy1 * f.y[0] * i11;
  gL[4] = -gy1 * f.y[1] * i11 - glp * i11;
  gy0 -= gy1 * L[3] * i11;
  gd[0] = gy0 * i00;
  gL[0] = -gy0 * f.y[0] * i00 - glp * i00;
  gL[1] = gL[2] = gL[5] = 0.f;
}

// ---- -logsumexp([lp + log .999, -log V + log .001])  (MixWithUniformProbability) -------------------
TTK_HD float mix_uniform_nll(float lp, float log_volume, float& dv_dlp) {
  const float a = lp + logf(0.999f), b = -log_volume + logf(0.001f);
  const float m = fmaxf(a, b);
  const float ea = expf(a - m), eb = expf(b - m);
  dv_dlp = -ea / (ea + eb);
  return -(m + logf(ea + eb));
}

// ---- rotation_delta(q, t) = to_rotvec(conj(q) * t) -------------------------------------------------
struct RotDelta {
  Q m;          // conj(q)*t after positivereal
  float s, nv, angle, r[3];
};
TTK_HD RotDelta rotation_delta(const float q[4], const float t[4]) {
  RotDelta o;
  Q m = hm::qmul(Q{-q[0], -q[1], -q[2], q[3]}, Q{t[0], t[1], t[2], t[3]});
  o.s = m.w > 0.f ? 1.f : (m.w < 0.f ? -1.f : 0.f);  // torch.sign
  m = Q{m.i * o.s, m.j * o.s, m.k * o.s, m.w * o.s};
  o.m = m;
  o.nv = sqrtf(m.i * m.i + m.j * m.j + m.k * m.k);
  o.angle = 2.f * atan2f(o.nv, m.w);
  const float f = o.angle / (o.nv + 1.0e-12f);
  o.r[0] = m.i * f; o.r[1] = m.j * f; o.r[2] = m.k * f;
  return o;
}
TTK_HD void rotation_delta_bwd(const float q[4], const float t[4], const RotDelta& o, const float gr[3], float gq[4]) {
  const float eps = 1.0e-12f;
  const Q m = o.m;
  const float den = o.nv + eps;
  const float f = o.angle / den;
  const float dot = gr[0] * m.i + gr[1] * m.j + gr[2] * m.k;
  const float n2 = o.nv * o.nv + m.w * m.w;
  const float dA_dnv = n2 > 0.f ? 2.f * m.w / n2 : 0.f;
  const float dA_dw = n2 > 0.f ? -2.f * o.nv / n2 : 0.f;
  const float df_dnv = dA_dnv / den - o.angle / (den * den);
  const float c = o.nv > 0.f ? dot * df_dnv / o.nv : 0.f;
  Q gm{gr[0] * f + c * m.i, gr[1] * f + c * m.j, gr[2] * f + c * m.k, dot * dA_dw / den};
  gm = Q{gm.i * o.s, gm.j * o.s, gm.k * o.s, gm.w * o.s};
  // m = conj(q) * t  ->  dL/dconj(q) = gm * conj(t)
  const Q gqc = hm::qmul(gm, Q{-t[0], -t[1], -t[2], t[3]});
  gq[0] = -gqc.i; gq[1] = -gqc.j; gq[2] = -gqc.k; gq[3] = gqc.w;
}

// QuatPoseNLLLoss: volume = 4/3 pi^4
TTK_HD float nllrot_loss(const float q[4], const float t[4], const float L[9]) {
  const RotDelta d = rotation_delta(q, t);
  const Mvn3 f = mvn_logprob(d.r, L);
  float dummy;
  return mix_uniform_nll(f.lp, logf(4.f / 3.f * 97.409091034002437236f), dummy);
}
TTK_HD void nllrot_loss_bwd(const float q[4], const float t[4], const float L[9], float gv, float gq[4], float gL[9]) {
  const RotDelta d = rotation_delta(q, t);
  const Mvn3 f = mvn_logprob(d.r, L);
  float dv_dlp;
  mix_uniform_nll(f.lp, logf(4.f / 3.f * 97.409091034002437236f), dv_dlp);
  float gd[3];
  mvn_logprob_bwd(L, f, dv_dlp * gv, gd, gL);
  rotation_delta_bwd(q, t, d, gd, gq);
}
// CorrelatedCoordPoseNLLLoss: MVN(coord, L).log_prob(target), volume 4
TTK_HD float nllcoord_loss(const float c[3], const float t[3], const float L[9]) {
  const float d[3] = {t[0] - c[0], t[1] - c[1], t[2] - c[2]};
  const Mvn3 f = mvn_logprob(d, L);
  float dummy;
  return mix_uniform_nll(f.lp, logf(4.f), dummy);
}
TTK_HD void nllcoord_loss_bwd(const float c[3], const float t[3], const float L[9], float gv, float gc[3], float gL[9]) {
  const float d[3] = {t[0] - c[0], t[1] - c[1], t[2] - c[2]};
  const Mvn3 f = mvn_logprob(d, L);
  float dv_dlp;
  mix_uniform_nll(f.lp, logf(4.f), dv_dlp);
  float gd[3];
  mvn_logprob_bwd(L, f, dv_dlp * gv, gd, gL);
  gc[0] = -gd[0]; gc[1] = -gd[1]; gc[2] = -gd[2];
}

// ---- smooth_geodesic_distance (losses.py:24-32): smooth_l1(|rotation_delta(q, t)|, 0, beta = 1 degree) / pi ---------------
TTK_HD float smooth_geodesic_loss(const float q[4], const float t[4]) {
  const RotDelta d = rotation_delta(q, t);
  const float th = sqrtf(d.r[0] * d.r[0] + d.r[1] * d.r[1] + d.r[2] * d.r[2]);
  const float beta = 3.14159265358979323846f / 180.f;
  return (th < beta ? 0.5f * th * th / beta : th - 0.5f * beta) / 3.14159265358979323846f;
}
TTK_HD void smooth_geodesic_loss_bwd(const float q[4], const float t[4], float gv, float gq[4]) {
  const RotDelta d = rotation_delta(q, t);
  const float th = sqrtf(d.r[0] * d.r[0] + d.r[1] * d.r[1] + d.r[2] * d.r[2]);
  const float beta = 3.14159265358979323846f / 180.f;
  const float gth = (th < beta ? th / beta : 1.f) / 3.14159265358979323846f * gv;
  const float f = th > 0.f ? gth / th : 0.f;  // d|r|/dr = r / |r| (0 at r = 0, torch's norm subgradient)
  const float gr[3] = {f * d.r[0], f * d.r[1], f * d.r[2]};
  rotation_delta_bwd(q, t, d, gr, gq);
}

// ---- elementwise distance kinds of LOSS_OBJECT_MAP (losses.py:16-21): MSELoss, L1Loss, SmoothL1Loss(beta) ------------------
enum { kElemL2 = 0, kElemL1 = 1, kElemSmoothL1 = 2 };
TTK_HD float elem_loss(int kind, float e, float beta) {
  const float a = fabsf(e);
  if (kind == kElemL2) return e * e;
  if (kind == kElemL1) return a;
  return a < beta ? 0.5f * e * e / beta : a - 0.5f * beta;
}
TTK_HD float elem_loss_d(int kind, float e, float beta) {
  const float sg = e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f);
  if (kind == kElemL2) return 2.f * e;
  if (kind == kElemL1) return sg;
  return fabsf(e) < beta ? e / beta : sg;
}

// ---- -Laplace(mu, b).log_prob(x), one element (negloglikelihood.py:68-69, DISTRIBUTION_CLASS_MAP["laplace"]) --------------
TTK_HD float laplace_nll(float mu, float b, float x) { return logf(2.f * b) + fabsf(x - mu) / b; }
TTK_HD void laplace_nll_bwd(float mu, float b, float x, float g, float& gmu, float& gb) {
  const float d = x - mu, ib = 1.f / b;
  const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  gmu = -sg * ib * g;
  gb = (ib - fabsf(d) * ib * ib) * g;
}

// ---- -Normal(mu, sigma).log_prob(x), one element ----------------------------------------------------
TTK_HD float normal_nll(float mu, float sigma, float x) {
  const float d = x - mu;
  return d * d / (2.f * sigma * sigma) + logf(sigma) + kHalfLog2Pi;
}
TTK_HD void normal_nll_bwd(float mu, float sigma, float x, float g, float& gmu, float& gsigma) {
  const float d = x - mu, is = 1.f / sigma;
  gmu = -d * is * is * g;
  gsigma = (-d * d * is * is * is + is) * g;
}

// ---- Points3dLoss / Points3dNLLLoss point weights (facemodel/keypoints68.py:79-80,106) -------------
TTK_HD float point_weight(int p, float chin, float eye) {
  if ((p >= 0 && p <= 7) || (p >= 9 && p <= 16)) return chin;
  if (p == 37 || p == 38 || p == 40 || p == 41 || p == 43 || p == 44 || p == 46 || p == 47) return eye;
  return 1.f;
}

// ---- ShapePlausibilityLoss: -logsumexp_k(ck - 0.5*sum_d((x-mu)*sinv)^2) * fudge, in float64 ---------
// ck = log w_k + sum_d log sinv_kd - 25 log 2pi (precomputed on the host in float64)
TTK_HD double gmm_nll(const float x[50], const double* ck, const double* mu, const double* sinv, int K, double fudge,
                      double* post /*[K] or null: softmax responsibilities*/) {
  double a[16];
  double mx = -1.0e300;
  for (int k = 0; k < K; ++k) {
    double e = 0.0;
    for (int d = 0; d < 50; ++d) {
      const double z = ((double)x[d] - mu[k * 50 + d]) * sinv[k * 50 + d];
      e += z * z;
    }
    a[k] = ck[k] - 0.5 * e;
    if (a[k] > mx) mx = a[k];
  }
  double s = 0.0;
  for (int k = 0; k < K; ++k) s += exp(a[k] - mx);
  if (post)
    for (int k = 0; k < K; ++k) post[k] = exp(a[k] - mx) / s;
  return -(mx + log(s)) * fudge;
}

// =====================================================================================================
// 6D-rotation variants (losses.py:53-64, torch6drotation.py:20-24,68-72, torchquaternion.py:70-168)
// =====================================================================================================
// torchquaternion.tomatrix: unit quaternion (i,j,k,w) -> row-major R
TTK_HD void quat_to_matrix(const float q[4], float R[9]) {
  const float qi = q[0], qj = q[1], qk = q[2], qw = q[3];
  R[0] = 1.f - 2.f * (qj * qj + qk * qk);
  R[3] = 2.f * (qi * qj + qk * qw);
  R[6] = 2.f * (qi * qk - qj * qw);
  R[1] = 2.f * (qi * qj - qk * qw);
  R[4] = 1.f - 2.f * (qi * qi + qk * qk);
  R[7] = 2.f * (qj * qk + qi * qw);
  R[2] = 2.f * (qi * qk + qj * qw);
  R[5] = 2.f * (qj * qk - qi * qw);
  R[8] = 1.f - 2.f * (qi * qi + qj * qj);
}
// Rot6dReprLoss: 0.75 - 0.25 * trace(R T^T), T = tomatrix(target quaternion)
TTK_HD float rot6d_loss(const float R[9], const float tq[4]) {
  float T[9], tr = 0.f;
  quat_to_matrix(tq, T);
  for (int i = 0; i < 9; ++i) tr += R[i] * T[i];
  return 0.75f - 0.25f * tr;
}
TTK_HD void rot6d_loss_bwd(const float tq[4], float gv, float gR[9]) {
  float T[9];
  quat_to_matrix(tq, T);
  for (int i = 0; i < 9; ++i) gR[i] = -0.25f * gv * T[i];
}
// Rot6dNormalizationSoftConstraint: mean((M M^T - I_2)^2), M = [x; y]
TTK_HD float ortho6d_loss(const float z[6]) {
  const float xx = hm::v3dot(z, z) - 1.f, yy = hm::v3dot(z + 3, z + 3) - 1.f, xy = hm::v3dot(z, z + 3);
  return 0.25f * (xx * xx + yy * yy + 2.f * xy * xy);
}
TTK_HD void ortho6d_loss_bwd(const float z[6], float gv, float gz[6]) {
  const float xx = hm::v3dot(z, z) - 1.f, yy = hm::v3dot(z + 3, z + 3) - 1.f, xy = hm::v3dot(z, z + 3);
  for (int i = 0; i < 3; ++i) {
    gz[i] = gv * (xx * z[i] + xy * z[3 + i]);
    gz[3 + i] = gv * (yy * z[3 + i] + xy * z[i]);
  }
}

// torchquaternion.from_matrix (Mat33Repr.as_quat): four candidate solutions, the best-conditioned one (largest
// square-root argument; ties -> first in the order k, j, i, w) is taken, then positivereal.  Table rows: the
// component whose square root is taken, the signs of (m00, m11, m22) in its argument, and for the other three
// components (dst, a, b, sign): q_dst = 0.25 * (m[a] + sign * m[b]) / q_main.
struct FromMatrixBranch {
  int main;
  float sd[3];
  int dst[3], a[3], b[3];
  float sg[3];
};
TTK_HD FromMatrixBranch from_matrix_branch(int pick) {
  // m index = 3*row + col
  switch (pick) {
    case 0: return {2, {-1.f, -1.f, 1.f}, {3, 0, 1}, {3, 6, 5}, {1, 2, 7}, {-1.f, 1.f, 1.f}};   // from k
    case 1: return {1, {-1.f, 1.f, -1.f}, {3, 0, 2}, {2, 3, 5}, {6, 1, 7}, {-1.f, 1.f, 1.f}};   // from j
    case 2: return {0, {1.f, -1.f, -1.f}, {3, 1, 2}, {7, 3, 2}, {5, 1, 6}, {-1.f, 1.f, 1.f}};   // from i
    default: return {3, {1.f, 1.f, 1.f}, {0, 1, 2}, {7, 2, 3}, {5, 6, 1}, {-1.f, -1.f, -1.f}};  // from w
  }
}
TTK_HD int from_matrix_pick(const float m[9], float args[4]) {
  const float d0 = m[0], d1 = m[4], d2 = m[8];
  args[0] = fmaxf(-d0 - d1 + d2 + 1.f, 1.0e-6f);
  args[1] = fmaxf(-d0 + d1 - d2 + 1.f, 1.0e-6f);
  args[2] = fmaxf(d0 - d1 - d2 + 1.f, 1.0e-6f);
  args[3] = fmaxf(d0 + d1 + d2 + 1.f, 1.0e-6f);
  int pick = 0;
  for (int i = 1; i < 4; ++i)
    if (args[i] > args[pick]) pick = i;
  return pick;
}
TTK_HD float sign0(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
TTK_HD void from_matrix(const float m[9], float q[4]) {
  float args[4];
  const int pick = from_matrix_pick(m, args);
  const FromMatrixBranch br = from_matrix_branch(pick);
  const float qm = 0.5f * sqrtf(args[pick]);
  q[br.main] = qm;
  for (int n = 0; n < 3; ++n) q[br.dst[n]] = 0.25f * (m[br.a[n]] + br.sg[n] * m[br.b[n]]) / qm;
  const float s = sign0(q[3]);
  for (int i = 0; i < 4; ++i) q[i] *= s;
}
TTK_HD void from_matrix_bwd(const float m[9], const float gq[4], float gm[9]) {
  for (int i = 0; i < 9; ++i) gm[i] = 0.f;
  float args[4], q[4];
  const int pick = from_matrix_pick(m, args);
  const FromMatrixBranch br = from_matrix_branch(pick);
  const float qm = 0.5f * sqrtf(args[pick]);
  q[br.main] = qm;
  for (int n = 0; n < 3; ++n) q[br.dst[n]] = 0.25f * (m[br.a[n]] + br.sg[n] * m[br.b[n]]) / qm;
  const float s = sign0(q[3]);
  float gmain = s * gq[br.main];
  for (int n = 0; n < 3; ++n) {
    const float g = s * gq[br.dst[n]];
    gmain -= g * q[br.dst[n]] / qm;
    const float f = 0.25f * g / qm;
    gm[br.a[n]] += f;
    gm[br.b[n]] += br.sg[n] * f;
  }
  const float raw = br.sd[0] * m[0] + br.sd[1] * m[4] + br.sd[2] * m[8] + 1.f;
  if (raw > 1.0e-6f) {  // clamp(min) passes no gradient below the bound
    const float gS = gmain * 0.125f / qm;
    gm[0] += br.sd[0] * gS;
    gm[4] += br.sd[1] * gS;
    gm[8] += br.sd[2] * gS;
  }
}

}  // namespace lm
}  // namespace ttk
