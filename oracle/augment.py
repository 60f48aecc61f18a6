"""CPU oracle of the crop augmentation: numpy float32 restatement of
GeneralFocusRoi._compute_view_roi / the crop transform / the label transforms / the bilinear warp.
TEST INFRASTRUCTURE; pinned by tests/golden/augment.npz (oracle/tools/gen_golden_augment.py).
Paths relative to /root/reference/trackertraincode/datatransformation/."""
from __future__ import annotations

import numpy as np

F = np.float32

FLIP_MAP = ([*range(16, -1, -1)] + [*range(26, 16, -1)] + [27, 28, 29, 30] + [35, 34, 33, 32, 31] + [45, 44, 43, 42, 47, 46]
            + [39, 38, 37, 36, 41, 40] + [54, 53, 52, 51, 50, 49, 48] + [59, 58, 57, 56, 55] + [64, 63, 62, 61, 60] + [67, 66, 65])


def compute_view_roi(face_bbox, f, t, bbs=0.3):
    """batch/geometric.py:108-157 in float32, same operation order (the INTEGER result must be bit-exact)."""
    bb, f, t = np.asarray(face_bbox, F), np.asarray(f, F), np.asarray(t, F)
    x0, y0, x1, y1 = bb[..., 0], bb[..., 1], bb[..., 2], bb[..., 3]
    rx, ry = t[..., 0], t[..., 1]
    w, h = x1 - x0, y1 - y0
    cx, cy = F(0.5) * (x1 + x0), F(0.5) * (y1 + y0)
    size = np.maximum(w, h) * f
    wx = F(0.5) * np.abs(size - w) + F(bbs) * np.minimum(size, w)
    wy = F(0.5) * np.abs(size - h) + F(bbs) * np.minimum(size, h)
    tx, ty = wx * rx, wy * ry
    return np.stack([cx - size * F(0.5) + tx, cy - size * F(0.5) + ty, cx + size * F(0.5) + tx, cy + size * F(0.5) + ty], -1).astype(F)


def round_view_roi(v):
    """torch.round(...).to(int32) (:205): round half to even."""
    return np.rint(v).astype(np.int32)


def crop_transform(view_roi, angle, N):
    """_center_rotation_tr(angle) @ _compute_point_transform_from_roi(view_roi) (:159-177) as (...,2,3)."""
    vr = np.asarray(view_roi, np.float64)
    sx, sy = N / (vr[..., 2] - vr[..., 0]), N / (vr[..., 3] - vr[..., 1])
    ox, oy = -vr[..., 0] * sx, -vr[..., 1] * sy
    c, s, h = np.cos(angle), np.sin(angle), 0.5 * N
    t0, t1 = h - (c * h - s * h), h - (s * h + c * h)
    m = np.stack([np.stack([c * sx, -s * sy, c * ox - s * oy + t0], -1), np.stack([s * sx, c * sy, s * ox + c * oy + t1], -1)], -2)
    return m.astype(F)


def _det(m):
    return m[..., 0, 0] * m[..., 1, 1] - m[..., 0, 1] * m[..., 1, 0]


def transform_coord(m, coord):  # tensors/affinetrafo.py:107-114
    xy = np.einsum("...ij,...j->...i", m[..., :, :2], coord[..., :2]) + m[..., :, 2]
    sc = np.sqrt((m[..., :, :2] ** 2).sum((-2, -1))) / np.sqrt(2.0)
    return np.concatenate([xy, (sc * coord[..., 2])[..., None]], -1).astype(F)


def _qmul(u, v):
    ui, uj, uk, uw = np.moveaxis(u, -1, 0)
    vi, vj, vk, vw = np.moveaxis(v, -1, 0)
    return np.stack([uw * vi + ui * vw + uj * vk - uk * vj, uw * vj - ui * vk + uj * vw + uk * vi,
                     uw * vk + ui * vj - uj * vi + uk * vw, uw * vw - ui * vi - uj * vj - uk * vk], -1)


def transform_rot(m, quat):  # :117-148
    sg = np.sign(_det(m))
    alpha = np.arctan2(-m[..., 0, 1], m[..., 1, 1])
    z = np.stack([np.zeros_like(alpha), np.zeros_like(alpha), np.sin(alpha / 2) * sg, np.cos(alpha / 2)], -1)
    out = _qmul(np.broadcast_to(z, quat.shape), quat)
    out[..., 1] *= sg
    out[..., 2] *= sg
    return out.astype(F)


def transform_points(m, pts):  # :37-64 (m: (2,3) or (B,2,3) with pts (B,68,3))
    mm = m[..., None, :, :] if pts.ndim == m.ndim else m
    xy = np.einsum("...ij,...j->...i", mm[..., :, :2], pts[..., :2]) + mm[..., :, 2]
    z = np.sqrt(np.abs(_det(m)))[..., None] * pts[..., 2]
    return np.concatenate([xy, z[..., None]], -1).astype(F)


def transform_keypoints(m, pts):  # :67-77
    out = transform_points(m, pts)
    det = _det(m)
    if det.ndim == 0:
        return out[FLIP_MAP] if det < 0 else out
    out = out.copy()
    for b in np.nonzero(det < 0)[0]:
        out[b] = out[b][FLIP_MAP]
    return out


def transform_roi(m, roi):  # :91-104
    x0, y0, x1, y1 = np.moveaxis(roi, -1, 0)
    corners = np.stack([np.stack([x0, y0], -1), np.stack([x0, y1], -1), np.stack([x1, y0], -1), np.stack([x1, y1], -1)], -2)
    mm = m[..., None, :, :]
    p = np.einsum("...ij,...j->...i", mm[..., :, :2], corners) + mm[..., :, 2]
    return np.concatenate([p.min(-2), p.max(-2)], -1).astype(F)


def normalization(N):  # position_normalization (:11-12)
    return np.array([[2.0 / N, 0, -1.0], [0, 2.0 / N, -1.0]], F)


def warp_bilinear(img, m, N):
    """affine_grid + grid_sample(bilinear, zeros, align_corners=False) of tensors/image_geometric_torch.py:60-98:
    out[i,j] = bilinear(img, m^-1 (j+.5, i+.5) - .5).  img (H,W) float; m (2,3)."""
    H, W = img.shape
    m = m.astype(np.float64)
    A, t = m[:, :2], m[:, 2]
    Ai = np.linalg.inv(A)
    jj, ii = np.meshgrid(np.arange(N) + 0.5, np.arange(N) + 0.5)
    p = np.stack([jj - t[0], ii - t[1]], -1) @ Ai.T
    u, v = p[..., 0] - 0.5, p[..., 1] - 0.5
    x0, y0 = np.floor(u).astype(int), np.floor(v).astype(int)
    ax, ay = u - x0, v - y0

    def at(y, x):
        ok = (x >= 0) & (x < W) & (y >= 0) & (y < H)
        return np.where(ok, img[np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)], 0.0)

    return ((at(y0, x0) * (1 - ax) + at(y0, x0 + 1) * ax) * (1 - ay) + (at(y0 + 1, x0) * (1 - ax) + at(y0 + 1, x0 + 1) * ax) * ay).astype(F)
