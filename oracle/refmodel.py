"""CPU oracle for the pose-estimator training step: a functional, pure-torch (fp32, CPU) restatement
of the reference's algorithm, each function citing the reference file:line it follows.

TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this module; the product (neuralnet-tracker-traincode_amd/) never does.

Pinning: checked against the golden vectors in tests/golden/ that were produced by importing the
reference itself (oracle/tools/gen_golden.py) - see tests/test_oracle_golden.py.

The state is a flat dict name -> tensor using the reference's state-dict key names
(SURVEY.md Appendix C); parameters are leaf tensors with requires_grad, buffers are updated in place.
All paths are relative to /root/reference/trackertraincode/.
"""
from __future__ import annotations

import math
from collections import defaultdict
from typing import Callable

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor

BN_EPS = 1.0e-5

# (name, cin, cout, stride) - backbones/mobilenet_v1.py:128-140
MOBILENET_BLOCKS = [
    ("dw2_1", 32, 64, 1), ("dw2_2", 64, 128, 2), ("dw3_1", 128, 128, 1), ("dw3_2", 128, 256, 2),
    ("dw4_1", 256, 256, 1), ("dw4_2", 256, 512, 2), ("dw5_1", 512, 512, 1), ("dw5_2", 512, 512, 1),
    ("dw5_3", 512, 512, 1), ("dw5_4", 512, 512, 1), ("dw5_5", 512, 512, 1), ("dw5_6", 512, 1024, 2),
    ("dw6", 1024, 1024, 1),
]
INTERMEDIATE_AFTER = ("dw2_1", "dw3_1", "dw4_1", "dw5_5", "dw6")  # mobilenet_v1.py:165-177


# =============================================================================================
# state handling
# =============================================================================================
def state_shapes(enable_point_head=True, enable_uncertainty=False, num_features=1024, enable_6drot=False, use_blurpool=False) -> dict:
    """Key -> shape inventory of NetworkWithPointHead("mobilenetv1") (SURVEY.md Appendix C;
    neuralnets/models.py:262-307, backbones/mobilenet_v1.py:122-140).  use_blurpool: the strided blocks hold
    conv_dw = Sequential(BlurPool2D, Conv2d) (mobilenet_v1.py:43-55): buffer conv_dw.0.kernel, weight conv_dw.1.weight."""
    s: dict[str, tuple] = {}

    def bn(prefix, c):
        s[prefix + ".weight"] = (c,)
        s[prefix + ".bias"] = (c,)
        s[prefix + ".running_mean"] = (c,)
        s[prefix + ".running_var"] = (c,)
        s[prefix + ".num_batches_tracked"] = ()

    s["convnet.conv1.weight"] = (32, 1, 5, 5)
    bn("convnet.bn1", 32)
    for name, cin, cout, stride in MOBILENET_BLOCKS:
        if use_blurpool and stride == 2:
            s[f"convnet.{name}.conv_dw.0.kernel"] = (3, 3)
            s[f"convnet.{name}.conv_dw.1.weight"] = (cin, 1, 3, 3)
        else:
            s[f"convnet.{name}.conv_dw.weight"] = (cin, 1, 3, 3)
        bn(f"convnet.{name}.bn_dw", cin)
        s[f"convnet.{name}.conv_sep.weight"] = (cout, cin, 1, 1)
        bn(f"convnet.{name}.bn_sep", cout)
    Fd = num_features

    def lin(prefix, o):
        s[prefix + ".weight"] = (o, Fd)
        s[prefix + ".bias"] = (o,)

    def tri(prefix):
        s[prefix + ".min_diag"] = (6,)
        lin(prefix + ".neck.lin", 7)

    lin("boxnet.linear", 4)
    if enable_uncertainty:
        s["boxnet.scales.hidden_scale"] = (5,)
    lin("posnet.linear_xy", 2)
    lin("posnet.linear_size", 1)
    if enable_uncertainty:
        tri("posnet.scales")
    lin("quatnet.linear", 6 if enable_6drot else 4)
    if enable_uncertainty:
        tri("quatnet.uncertainty_net")
    s["local_pose_offset.p"] = (8, 4)
    s["local_pose_offset_kpts.p"] = (8, 4)
    if enable_point_head:
        s["landmarks.deformablekeypoints.keypts"] = (68, 3)
        s["landmarks.deformablekeypoints.keyeigvecs"] = (50, 68, 3)
        lin("landmarks.shapenet", 50)
        if enable_uncertainty:
            s["landmarks.point_distrib_scales.hidden_scale"] = (69,)
            s["landmarks.shape_distrib_scales.hidden_scale"] = (51,)
    return s


BUFFER_LEAVES = ("running_mean", "running_var", "num_batches_tracked", "min_diag", "keypts", "keyeigvecs", "kernel")


def is_buffer(key: str) -> bool:
    return key.rsplit(".", 1)[-1] in BUFFER_LEAVES


def state_from_numpy(sd: dict, requires_grad=True, device="cpu") -> dict[str, Tensor]:
    out = {}
    for k, v in sd.items():
        t = torch.as_tensor(np.array(v)).to(device)
        if requires_grad and not is_buffer(k):
            t.requires_grad_(True)
        out[k] = t
    return out


def variance_param_keys(state) -> list[str]:
    """Parameters of FeaturesAsTriangularScale / DiagonalScaleParameter modules
    (scripts/train_poseestimator.py:114-122): these train at 0.1 x lr."""
    return [
        k for k in state
        if not is_buffer(k) and (".scales." in k or ".uncertainty_net." in k or "_distrib_scales." in k)
    ]


# =============================================================================================
# small math (neuralnets/math.py, torchquaternion.py) - quaternion order (i, j, k, w)
# =============================================================================================
def smoothclip0(x: Tensor) -> Tensor:
    """math.py:34-37: elu(x) + 1"""
    return F.elu(x) + 1.0


def qmul(u: Tensor, v: Tensor) -> Tensor:
    """Hamilton product, real part last (torchquaternion.py:23-48)."""
    ui, uj, uk, uw = u.unbind(-1)
    vi, vj, vk, vw = v.unbind(-1)
    return torch.stack(
        [
            ui * vw + uw * vi - uk * vj + uj * vk,
            uj * vw + uk * vi + uw * vj - ui * vk,
            uk * vw - uj * vi + ui * vj + uw * vk,
            uw * vw - ui * vi - uj * vj - uk * vk,
        ],
        dim=-1,
    )


def qconj(q: Tensor) -> Tensor:
    return q * q.new_tensor([-1.0, -1.0, -1.0, 1.0])


def qrotate(q: Tensor, p: Tensor) -> Tensor:
    """(q * p) * conj(q) with p a pure-imaginary quaternion (torchquaternion.py:51-67).  Not the
    unit-quaternion shortcut: the reference's form scales with |q|^2, which matters for d/dq."""
    pq = torch.cat([p, torch.zeros_like(p[..., :1])], dim=-1)
    return qmul(qmul(q, pq), qconj(q))[..., :3]


def qnormalize(q: Tensor) -> Tensor:
    """torchquaternion.py:221-222: F.normalize(p=2, eps=1e-6) = q / max(|q|, eps)"""
    n = q.norm(dim=-1, keepdim=True).clamp_min(1.0e-6)
    return q / n


def positivereal(q: Tensor) -> Tensor:
    """torchquaternion.py:216-218 (torch.sign: sign(0) = 0)"""
    return q * torch.sign(q[..., 3:])


def to_rotvec(q: Tensor, eps=1.0e-12) -> Tensor:
    """torchquaternion.py:187-199"""
    q = positivereal(q)
    v, w = q[..., :3], q[..., 3]
    n = v.norm(dim=-1, keepdim=True)
    angle = 2.0 * torch.atan2(n[..., 0], w)
    return v * angle[..., None] / (n + eps)


def rotation_delta(a: Tensor, b: Tensor) -> Tensor:
    """torchquaternion.py:202-206: rotvec(a^-1 * b)"""
    return to_rotvec(qmul(qconj(a), b))


def quat_to_matrix(q: Tensor) -> Tensor:
    """torchquaternion.py:70-91"""
    i, j, k, w = q.unbind(-1)
    rows = [
        1.0 - 2.0 * (j * j + k * k), 2.0 * (i * j - k * w), 2.0 * (i * k + j * w),
        2.0 * (i * j + k * w), 1.0 - 2.0 * (i * i + k * k), 2.0 * (j * k - i * w),
        2.0 * (i * k - j * w), 2.0 * (j * k + i * w), 1.0 - 2.0 * (i * i + j * j),
    ]
    return torch.stack(rows, dim=-1).view(*q.shape[:-1], 3, 3)


# =============================================================================================
# backbone  (backbones/mobilenet_v1.py:75-92, 160-186)
# =============================================================================================
def rot6d_to_matrix(z: Tensor) -> Tensor:
    """torch6drotation.tomatrix (:27-49): Gram-Schmidt by cross products, rows normalised with eps 1e-6, identity
    where max|R R^T - I| > 1e-3 (no gradient through the replaced samples)."""
    x, y = z[..., :3], z[..., 3:]
    c = torch.cross(x, y, dim=-1)
    y2 = torch.cross(c, x, dim=-1)
    out = torch.nn.functional.normalize(torch.stack([x, y2, c], dim=-2), dim=-1, eps=1e-6)
    eye = torch.eye(3, dtype=z.dtype)
    bad = (out @ out.transpose(-2, -1) - eye).flatten(-2).abs().amax(-1)
    return torch.where(bad[..., None, None] > 1.0e-3, eye, out)


def matrix_to_quat(m: Tensor) -> Tensor:
    """torchquaternion.from_matrix (:94-168) = Mat33Repr.as_quat: four candidate solutions, the one with the largest
    square-root argument (ties: first of k, j, i, w) is picked without gradient, then positivereal."""
    d0, d1, d2 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
    args = torch.stack([-d0 - d1 + d2, -d0 + d1 - d2, d0 - d1 - d2, d0 + d1 + d2], dim=-1) + 1.0
    args = torch.clamp(args, 1.0e-6, None)
    qx = torch.sqrt(args) * 0.5
    qk, qj, qi, qw = qx.unbind(-1)
    mm = lambda a, b: m[..., a, b]
    cand = torch.stack([
        torch.stack([0.25 * (mm(2, 0) + mm(0, 2)) / qk, 0.25 * (mm(1, 2) + mm(2, 1)) / qk, qk, 0.25 * (mm(1, 0) - mm(0, 1)) / qk], -1),
        torch.stack([0.25 * (mm(1, 0) + mm(0, 1)) / qj, qj, 0.25 * (mm(1, 2) + mm(2, 1)) / qj, 0.25 * (mm(0, 2) - mm(2, 0)) / qj], -1),
        torch.stack([qi, 0.25 * (mm(1, 0) + mm(0, 1)) / qi, 0.25 * (mm(0, 2) + mm(2, 0)) / qi, 0.25 * (mm(2, 1) - mm(1, 2)) / qi], -1),
        torch.stack([0.25 * (mm(2, 1) - mm(1, 2)) / qw, 0.25 * (mm(0, 2) - mm(2, 0)) / qw, 0.25 * (mm(1, 0) - mm(0, 1)) / qw, qw], -1),
    ], dim=-2)
    with torch.no_grad():
        pick = torch.argmax(args, dim=-1)
    q = torch.gather(cand, -2, pick[..., None, None].expand(*pick.shape, 1, 4)).squeeze(-2)
    return positivereal(q)


def local_pose_offset_m(P: Tensor, R: Tensor, coord: Tensor, set_id: Tensor | None):
    """LocalToGlobalCoordinateOffset with Mat33Repr (modelcomponents.py:136-184; rotrepr.py:73-85: make_rotate_x
    takes the FULL angle for matrices)."""
    p = P[:1] if set_id is None else P[set_id.long()]
    ang = p[:, 1]
    sn, cs = torch.sin(ang), torch.cos(ang)
    one, zero = torch.ones_like(ang), torch.zeros_like(ang)
    Rx = torch.stack([one, zero, zero, zero, cs, -sn, zero, sn, cs], dim=-1).view(-1, 3, 3)
    transl = torch.cat([torch.zeros_like(p[:, :1]), p[:, 1:3]], dim=-1)
    scale = coord[..., 2:] * smoothclip0(p[:, 3])[..., None]
    Rn = R @ Rx
    pos = (R @ transl[..., :, None]).squeeze(-1)[..., :2] * scale + coord[..., :2]
    return Rn, torch.cat([pos, scale], dim=-1)


def _bn(y: Tensor, st, prefix: str, training: bool, momentum: float) -> Tensor:
    rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
    out = F.batch_norm(y, rm, rv, st[prefix + ".weight"], st[prefix + ".bias"], training, momentum, BN_EPS)
    if training:
        st[prefix + ".num_batches_tracked"] += 1
    return out


def mobilenet_forward(st, x: Tensor, training: bool, momentum: float = 0.1, prefix="convnet.",
                      want_raw: dict | None = None):
    """Returns (features[B,1024], [z65, z33, z17, z9, z5]).  `want_raw`, if given, receives every raw
    conv output (pre-BN) keyed by layer name - used by the per-layer GPU parity tests."""
    p = prefix
    y = F.conv2d(x, st[p + "conv1.weight"], None, stride=2, padding=2)  # mobilenet_v1.py:122-124,161
    if want_raw is not None:
        want_raw["conv1"] = y
    a = F.relu(_bn(y, st, p + "bn1", training, momentum))
    inter = []
    for name, cin, cout, stride in MOBILENET_BLOCKS:
        b = p + name
        if b + ".conv_dw.0.kernel" in st:
            # use_blurpool (mobilenet_v1.py:43-55; modelcomponents.py:187-205): BlurPool2D = kornia's _blur_pool_by_kernel2d
            # (kornia is a requirements.txt dependency without a pinned version, absent from this image; its published form:
            # conv2d with the binomial kernel repeated per channel, zero padding (k-1)//2, the stride, groups = C), then the
            # depthwise conv at stride 1
            t = F.conv2d(a, st[b + ".conv_dw.0.kernel"].repeat(cin, 1, 1, 1), None, stride=stride, padding=1, groups=cin)
            if want_raw is not None:
                want_raw[name + ".blur"] = t
            y = F.conv2d(t, st[b + ".conv_dw.1.weight"], None, stride=1, padding=1, groups=cin)
        else:
            y = F.conv2d(a, st[b + ".conv_dw.weight"], None, stride=stride, padding=1, groups=cin)
        if want_raw is not None:
            want_raw[name + ".dw"] = y
        h = F.relu(_bn(y, st, b + ".bn_dw", training, momentum))
        y = F.conv2d(h, st[b + ".conv_sep.weight"], None)
        if want_raw is not None:
            want_raw[name + ".pw"] = y
        o = _bn(y, st, b + ".bn_sep", training, momentum)
        if stride == 1 and cin == cout:  # mobilenet_v1.py:70,86-88
            o = o + a
        a = F.relu(o)
        if name in INTERMEDIATE_AFTER:
            inter.append(a)
    feat = a.mean(dim=(2, 3))  # AdaptiveAvgPool2d(1) + view, mobilenet_v1.py:143,180-181
    return feat, inter


# =============================================================================================
# heads  (neuralnets/models.py:96-215, negloglikelihood.py:22-65,187-242, modelcomponents.py:136-184)
# =============================================================================================
# =============================================================================================
# ResNet18 variant (backbones/resnet.py:52-104).  PARITY UNPINNED: the arithmetic lives in torchvision.models.resnet
# (requirements.txt:3, version not pinned; absent from this image), so this restates torchvision's published
# BasicBlock / ResNet.forward and cannot be checked against an import of the reference.  Anchors: the reference's call
# site (resnet.py:58-73: _resnet(BasicBlock, [2,2,2,2]), conv1 := Conv2d(1,64,7,2,3), children()[:-1] + Flatten) and
# its shape tests (test/test_backbones.py:19-40).
# =============================================================================================
RESNET18_PLAN = [(64, 1), (64, 1), (128, 2), (128, 1), (256, 2), (256, 1), (512, 2), (512, 1)]


def resnet18_state_shapes(prefix="", use_blurpool=False) -> dict:
    """use_blurpool (resnet.py:31-49,63-66): the reference's CustomBlock replaces conv1 by Sequential(BlurPool2D(stride), conv3x3 stride 1)
    in every block and the max-pool by BlurPool2D(3, channels 64, stride 2): buffers `kernel` [3, 3]."""
    s = {}

    def bn(name, c):
        s[name + ".weight"], s[name + ".bias"] = (c,), (c,)
        s[name + ".running_mean"], s[name + ".running_var"], s[name + ".num_batches_tracked"] = (c,), (c,), ()

    s[prefix + "layers.0.weight"] = (64, 1, 7, 7)
    bn(prefix + "layers.1", 64)
    if use_blurpool:
        s[prefix + "layers.3.kernel"] = (3, 3)
    cin = 64
    for i, (planes, stride) in enumerate(RESNET18_PLAN):
        b = f"{prefix}layers.{4 + i // 2}.{i % 2}"
        if use_blurpool:
            s[b + ".conv1.0.kernel"] = (3, 3)
            s[b + ".conv1.1.weight"] = (planes, cin, 3, 3)
        else:
            s[b + ".conv1.weight"] = (planes, cin, 3, 3)
        bn(b + ".bn1", planes)
        s[b + ".conv2.weight"] = (planes, planes, 3, 3)
        bn(b + ".bn2", planes)
        if stride != 1 or cin != planes:
            s[b + ".downsample.0.weight"] = (planes, cin, 1, 1)
            bn(b + ".downsample.1", planes)
        cin = planes
    return s


def resnet18_forward(st, x: Tensor, training: bool, momentum: float = 0.1, prefix=""):
    """[B,1,H,W] -> ([B,512], None)"""
    def blur(a, key, stride):  # BlurPool2D (modelcomponents.py:187-205; kornia's _blur_pool_by_kernel2d restated as in mobilenet_forward)
        c = a.shape[1]
        return F.conv2d(a, st[key].to(a.dtype).repeat(c, 1, 1, 1), None, stride=stride, padding=1, groups=c)

    y = F.conv2d(x, st[prefix + "layers.0.weight"], stride=2, padding=3)
    y = torch.relu(_bn(y, st, prefix + "layers.1", training, momentum))
    y = blur(y, prefix + "layers.3.kernel", 2) if prefix + "layers.3.kernel" in st else F.max_pool2d(y, 3, 2, 1)
    cin = 64
    for i, (planes, stride) in enumerate(RESNET18_PLAN):
        b = f"{prefix}layers.{4 + i // 2}.{i % 2}"
        identity = y
        if b + ".conv1.0.kernel" in st:  # CustomBlock: blur with the block's stride, then the convolution at stride 1
            out = F.conv2d(blur(y, b + ".conv1.0.kernel", stride), st[b + ".conv1.1.weight"], stride=1, padding=1)
        else:
            out = F.conv2d(y, st[b + ".conv1.weight"], stride=stride, padding=1)
        out = torch.relu(_bn(out, st, b + ".bn1", training, momentum))
        out = _bn(F.conv2d(out, st[b + ".conv2.weight"], stride=1, padding=1), st, b + ".bn2", training, momentum)
        if stride != 1 or cin != planes:
            identity = _bn(F.conv2d(y, st[b + ".downsample.0.weight"], stride=stride), st, b + ".downsample.1", training, momentum)
        y = torch.relu(out + identity)
        cin = planes
    return y.mean(dim=(2, 3)), None


def _linear(st, prefix, f):
    return F.linear(f, st[prefix + ".weight"], st[prefix + ".bias"])


def diagonal_scale_parameter(h: Tensor) -> Tensor:
    """negloglikelihood.py:50-65"""
    return smoothclip0(h[:1]) * smoothclip0(h[1:]) + 1.0e-6


def features_as_triangular_scale(st, prefix, f) -> Tensor:
    """negloglikelihood.py:22-35 (Neck), :187-211 (_fill_triangular_matrix), :214-242"""
    x = _linear(st, prefix + ".neck.lin", f)
    mult = smoothclip0(x[..., :1])
    v = x[..., 1:]
    z = torch.cat([smoothclip0(v[..., :3]), v[..., 3:]], dim=-1)
    z = mult * z + st[prefix + ".min_diag"]
    zero = torch.zeros_like(z[..., 0])
    rows = [z[..., 0], zero, zero, z[..., 3], z[..., 1], zero, z[..., 4], z[..., 5], z[..., 2]]
    return torch.stack(rows, dim=-1).view(*z.shape[:-1], 3, 3)


def local_pose_offset(P: Tensor, q: Tensor, coord: Tensor, set_id: Tensor | None):
    """modelcomponents.py:136-184.  Quirk kept: p[:,1] is BOTH the x-rotation angle and the first
    translation component, p[:,0] is unused (modelcomponents.py:146-156)."""
    p = P[:1] if set_id is None else P[set_id.long()]
    half = 0.5 * p[:, 1]
    zeros = torch.zeros_like(half)
    q_off = torch.stack([torch.sin(half), zeros, zeros, torch.cos(half)], dim=-1)
    transl = torch.stack([zeros, p[:, 1], p[:, 2]], dim=-1)
    s_off = smoothclip0(p[:, 3])
    size = coord[..., 2:] * s_off[..., None]
    q_new = qmul(q, q_off)
    corr = qrotate(q, transl)[..., :2] * size
    return q_new, torch.cat([corr + coord[..., :2], size], dim=-1)


def heads_forward(st, f: Tensor, set_id: Tensor | None, *, enable_point_head: bool,
                  enable_uncertainty: bool, use_local_pose_offset: bool = True, training: bool = True,
                  enable_6drot: bool = False):
    """neuralnets/models.py:340-376 after the backbone; `rot` is returned as a plain [B,4] quaternion
    ([B,3,3] matrix with the 6D head, models.py:153-174)."""
    out = {}
    z = _linear(st, "boxnet.linear", f)  # models.py:186-197
    size = smoothclip0(z[..., 2:])
    out["roi"] = torch.cat([z[..., :2] - size, z[..., :2] + size], dim=-1)
    if enable_uncertainty:
        out["roi_scales"] = diagonal_scale_parameter(st["boxnet.scales.hidden_scale"])[None, :].expand_as(z)
    coord = torch.cat([_linear(st, "posnet.linear_xy", f), smoothclip0(_linear(st, "posnet.linear_size", f))], -1)
    if enable_uncertainty:
        out["coord_scales"] = features_as_triangular_scale(st, "posnet.scales", f)
    zq = _linear(st, "quatnet.linear", f)  # models.py:135-150, rotrepr.py:36-48
    if enable_6drot:
        q = rot6d_to_matrix(zq)
        out["unnormalized_6drepr"] = zq
        offset = local_pose_offset_m
    else:
        qu = torch.cat([zq[..., :3], smoothclip0(zq[..., 3:])], dim=-1)
        q = qnormalize(qu)
        out["unnormalized_quat"] = qu
        offset = local_pose_offset
    if enable_uncertainty:
        out["pose_scales_tril"] = features_as_triangular_scale(st, "quatnet.uncertainty_net", f)
    hidden_q, hidden_c = q, coord
    if use_local_pose_offset:
        q, coord = offset(st["local_pose_offset.p"], hidden_q, hidden_c, set_id)
    out["rot"], out["coord"] = q, coord
    if enable_point_head:
        qk, ck = q, coord
        if use_local_pose_offset:
            qk, ck = offset(st["local_pose_offset_kpts.p"], hidden_q, hidden_c, set_id)
        shp = _linear(st, "landmarks.shapenet", f)  # models.py:108-124
        eig = st["landmarks.deformablekeypoints.keyeigvecs"]
        local = (eig[None] * shp[:, :, None, None]).sum(dim=1) + st["landmarks.deformablekeypoints.keypts"]
        rotated = (qk[:, None] @ local[..., None]).squeeze(-1) if enable_6drot else qrotate(qk[:, None, :], local)
        pts = rotated * ck[:, None, 2:]  # modelcomponents.py:38-56
        pts = torch.cat([pts[..., :2] + ck[:, None, :2], pts[..., 2:]], dim=-1)
        out["pt3d_68"], out["shapeparam"] = pts, shp
        if enable_uncertainty:
            ps = diagonal_scale_parameter(st["landmarks.point_distrib_scales.hidden_scale"])
            ss = diagonal_scale_parameter(st["landmarks.shape_distrib_scales.hidden_scale"])
            out["pt3d_68_scales"] = ps[None, :, None].expand_as(pts)
            out["shapeparam_scales"] = ss[None, :].expand_as(shp)
    if not training:
        out["pose"] = matrix_to_quat(out["rot"]) if enable_6drot else out["rot"]
    return out


def network_forward(st, x: Tensor, set_id: Tensor | None, cfg: dict, training: bool, momentum=0.1):
    assert x.shape[2] == 129 and x.shape[3] == 129  # models.py:341
    if cfg.get("config", "mobilenetv1") == "resnet18":  # models.py:221-222 (create_pose_estimator_backbone): ResNetBackbone, 512 features
        f, _ = resnet18_forward(st, x, training, momentum, prefix="convnet.")
    else:
        f, _ = mobilenet_forward(st, x, training, momentum)
    return heads_forward(
        st, f, set_id, enable_point_head=cfg["enable_point_head"],
        enable_uncertainty=cfg["enable_uncertainty"],
        use_local_pose_offset=cfg.get("use_local_pose_offset", True), training=training,
        enable_6drot=cfg.get("enable_6drot", False),
    ), f


# =============================================================================================
# losses  (neuralnets/losses.py, negloglikelihood.py)
# =============================================================================================
def point_weights(chin=0.8, eye=0.0) -> Tensor:
    """losses.py:139-142 / facemodel/keypoints68.py:79-80,106"""
    w = torch.ones(68)
    w[list(range(0, 8))] = chin   # chin_left[:-1]
    w[list(range(9, 17))] = chin  # chin_right[1:]
    w[[37, 38, 41, 40, 43, 44, 47, 46]] = eye
    return w


def loss_rot(p, s):  # losses.py:42-50, torchquaternion.py:225-228
    return 1.0 - (p["rot"] * s["pose"]).sum(-1).square()


def loss_rot6d(p, s):  # losses.py:53-58, torch6drotation.py:68-72
    return 0.75 - 0.25 * (p["rot"] * quat_to_matrix(s["pose"])).flatten(-2).sum(-1)


def loss_ortho6d(p, s):  # losses.py:61-64, torch6drotation.py:20-24
    m = p["unnormalized_6drepr"].unflatten(-1, (2, 3))
    return (m @ m.transpose(-2, -1) - torch.eye(2, dtype=m.dtype)).square().flatten(-2).mean(-1)


def loss_xy(p, s):  # losses.py:79-88
    return (p["coord"][..., :2] - s["coord"][..., :2]).square().mean(-1)


def loss_sz(p, s):  # losses.py:67-76
    return (p["coord"][..., 2] - s["coord"][..., 2]).square()


def loss_box(p, s):  # losses.py:163-173
    return (p["roi"] - s["roi"]).square().mean(-1)


def loss_points3d(p, s, dim=3):  # losses.py:128-160
    d = (p["pt3d_68"][..., :dim] - s["pt3d_68"][..., :dim]).square().sum(-1)
    return (d * point_weights().to(d.device)[None, :]).mean(-1)


def loss_shp_l2(p, s):  # losses.py:91-97
    return (p["shapeparam"] - s["shapeparam"]).square().mean(-1)


def loss_quatreg(p, s):  # losses.py:116-125
    return (1.0 - p["unnormalized_quat"].norm(dim=1)).square()


class ShapeGmm:
    """losses.py:100-113 + modelcomponents.py:218-290; float64, diagonal covariances."""

    def __init__(self, npz_path: str):
        d = np.load(npz_path)
        self.w = torch.from_numpy(d["weights"])
        self.mu = torch.from_numpy(d["means"])
        self.sinv = torch.from_numpy(d["cov"]).rsqrt()
        self.normc = 0.5 * self.mu.shape[-1] * math.log(2 * math.pi)
        self.fudge = 0.001 / self.w.shape[0]

    def __call__(self, p, s):
        x = p["shapeparam"].to(torch.float64)
        dev = x.device
        delta = x[:, None, :] - self.mu.to(dev)
        e = -0.5 * (delta * self.sinv.to(dev)).square().sum(-1)
        nrm = torch.log(self.sinv.to(dev)).sum(-1) - self.normc
        ll = torch.logsumexp(torch.log(self.w.to(dev)) + e + nrm, dim=-1)
        return (-ll * self.fudge).to(torch.float32)


def mvn_tril_logprob(delta: Tensor, L: Tensor) -> Tensor:
    """log N(delta; 0, L L^T) for 3x3 lower-triangular L
    (torch.distributions.MultivariateNormal.log_prob as used at negloglikelihood.py:123-125,249-259)."""
    y0 = delta[..., 0] / L[..., 0, 0]
    y1 = (delta[..., 1] - L[..., 1, 0] * y0) / L[..., 1, 1]
    y2 = (delta[..., 2] - L[..., 2, 0] * y0 - L[..., 2, 1] * y1) / L[..., 2, 2]
    maha = y0 * y0 + y1 * y1 + y2 * y2
    half_log_det = L[..., 0, 0].log() + L[..., 1, 1].log() + L[..., 2, 2].log()
    return -0.5 * (3 * math.log(2 * math.pi) + maha) - half_log_det


def mix_with_uniform(lp: Tensor, volume: float) -> Tensor:
    """negloglikelihood.py:100-110"""
    a = lp + math.log(0.999)
    b = torch.full_like(lp, -math.log(volume) + math.log(0.001))
    return torch.logsumexp(torch.stack([a, b], dim=-1), dim=-1)


def loss_nllrot(p, s):  # negloglikelihood.py:245-274 (rot.as_quat(): from_matrix for the 6D head)
    q = matrix_to_quat(p["rot"]) if p["rot"].dim() == 3 else p["rot"]
    r = rotation_delta(q, s["pose"])
    return -mix_with_uniform(mvn_tril_logprob(r, p["pose_scales_tril"]), math.pi**4 * 4.0 / 3.0)


def loss_nllcoord(p, s):  # negloglikelihood.py:113-126
    return -mix_with_uniform(mvn_tril_logprob(s["coord"] - p["coord"], p["coord_scales"]), 4.0)


def _normal_logprob(x, mu, sigma):
    return -((x - mu) ** 2) / (2 * sigma * sigma) - sigma.log() - 0.5 * math.log(2 * math.pi)


def loss_nllbox(p, s):  # negloglikelihood.py:129-142
    return -_normal_logprob(s["roi"], p["roi"], p["roi_scales"]).mean(-1)


def loss_nllpoints3d(p, s, dim=3):  # negloglikelihood.py:145-166
    lp = _normal_logprob(s["pt3d_68"][..., :dim], p["pt3d_68"][..., :dim], p["pt3d_68_scales"][..., :dim])
    return (-point_weights().to(lp.device)[None, :, None] * lp).mean(dim=(-2, -1))


# ---- the non-default kinds of the loss switches (unused by the training script; pinned by tests/golden/loss_kinds.npz) ----------------
def elem_distance(kind: str, p: Tensor, t: Tensor) -> Tensor:
    """LOSS_OBJECT_MAP, losses.py:16-21: MSELoss / L1Loss / SmoothL1Loss(beta=0.01), reduction="none"."""
    e = p - t
    if kind == "l2":
        return e.square()
    if kind == "l1":
        return e.abs()
    assert kind == "smooth_l1"
    beta = 0.01
    return torch.where(e.abs() < beta, 0.5 * e.square() / beta, e.abs() - 0.5 * beta)


def loss_xy_kind(p, s, kind):  # losses.py:79-88
    return elem_distance(kind, p["coord"][..., :2], s["coord"][..., :2]).mean(-1)


def loss_sz_kind(p, s, kind):  # losses.py:67-76
    return elem_distance(kind, p["coord"][..., 2], s["coord"][..., 2])


def loss_box_kind(p, s, kind):  # losses.py:163-173
    return elem_distance(kind, p["roi"], s["roi"]).mean(-1)


def loss_points3d_kind(p, s, kind, dim=3, chin=0.8, eye=0.0):  # losses.py:128-160
    d = elem_distance(kind, p["pt3d_68"][..., :dim], s["pt3d_68"][..., :dim]).sum(-1)
    return (d * point_weights(chin, eye).to(d.device)[None, :]).mean(-1)


def loss_rot_smooth_geodesic(p, s):  # losses.py:24-32: smooth_l1(|rotation_delta|, 0, beta = 1 degree) / pi; torchquaternion.py:233-236
    th = rotation_delta(p["rot"], s["pose"]).norm(dim=-1)
    beta = math.pi / 180.0
    return torch.where(th < beta, 0.5 * th.square() / beta, th - 0.5 * beta) / math.pi


def _laplace_logprob(x, mu, b):  # torch.distributions.Laplace.log_prob
    return -(2 * b).log() - (x - mu).abs() / b


def _dist_logprob(distribution):
    return {"gaussian": _normal_logprob, "laplace": _laplace_logprob}[distribution]


def loss_nllcoord_indep(p, s, xy_weight, size_weight, distribution="gaussian"):  # negloglikelihood.py:72-97
    w = torch.tensor([xy_weight / 2.0, xy_weight / 2.0, size_weight], dtype=p["coord"].dtype)
    return -(_dist_logprob(distribution)(s["coord"], p["coord"], p["coord_scales"]) * w[None, :]).mean(-1)


def loss_nllbox_dist(p, s, distribution):  # negloglikelihood.py:129-142
    return -_dist_logprob(distribution)(s["roi"], p["roi"], p["roi_scales"]).mean(-1)


def loss_nllpoints3d_dist(p, s, distribution, dim=3, chin=0.8, eye=0.0):  # negloglikelihood.py:145-166
    lp = _dist_logprob(distribution)(s["pt3d_68"][..., :dim], p["pt3d_68"][..., :dim], p["pt3d_68_scales"][..., :dim])
    return (-point_weights(chin, eye).to(lp.device)[None, :, None] * lp).mean(dim=(-2, -1))


def loss_nllshape_dist(p, s, distribution):  # negloglikelihood.py:169-177
    return -_dist_logprob(distribution)(s["shapeparam"], p["shapeparam"], p["shapeparam_scales"]).mean(-1)


# ---------------------------------------------------------------------------------------------
# criterion tables  (scripts/train_poseestimator.py:170-285)
# ---------------------------------------------------------------------------------------------
def setup_losses(*, with_pointhead=True, with_nll_loss=False, rampup_nll_losses=False, epochs=200,
                 with_roi_train=True, gmm: Callable | None = None, enable_6drot=False):
    """Returns {tag_name: [(name, fn, weight or weight(epoch))]} for the train criterions."""

    def ramp(mult):
        if rampup_nll_losses:
            return lambda step: 0.01 * min(1.0, max(0.0, (step / epochs - 0.1) * 10.0)) * mult
        return mult * 0.01

    pose, roi, pts, pts25, shp = [], [], [], [], []
    # scripts/train_poseestimator.py:170-178: the 6D head swaps the rotation loss and the soft constraint, names stay
    reg = [("quatregularization1", loss_ortho6d if enable_6drot else loss_quatreg, 1.0e-6)]
    if with_nll_loss:
        pose += [("nllrot", loss_nllrot, ramp(0.5)), ("nllcoord", loss_nllcoord, ramp(0.5))]
        if with_roi_train:
            roi += [("nllbox", loss_nllbox, ramp(0.01))]
        if with_pointhead:
            pts += [("nllpoints3d", loss_nllpoints3d, ramp(0.5))]
            pts25 += [("nllpoints3d", lambda p, s: loss_nllpoints3d(p, s, 2), ramp(0.5))]
    pose += [("rot", loss_rot6d if enable_6drot else loss_rot, 1.0), ("xy", loss_xy, 0.25), ("sz", loss_sz, 0.25)]
    if with_roi_train:
        roi += [("box", loss_box, 0.01)]
    if with_pointhead:
        pts += [("points3d", loss_points3d, 0.5)]
        pts25 += [("points3d", lambda p, s: loss_points3d(p, s, 2), 0.5)]
        shp += [("shp_l2", loss_shp_l2, 0.1)]
        assert gmm is not None
        reg += [("nll_shp_gmm", gmm, 0.1)]
    train = {
        "ONLY_POSE": pose + reg + roi,
        "POSE_WITH_LMKS_NO_SHAPE_PARAMS": pose + reg + pts + roi,
        "POSE_WITH_LANDMARKS": pose + reg + pts + shp + roi,
        "POSE_WITH_LANDMARKS_3D_AND_2D": pose + reg + pts + shp + roi,
        "ONLY_LANDMARKS": pts + reg,
        "ONLY_LANDMARKS_25D": pts25 + reg,
    }
    test = {"POSE_WITH_LANDMARKS": pose + pts + roi + shp + reg}
    return train, test


def compute_loss(preds: dict, batches: list[dict], epoch: int, criterions: dict):
    """trackertraincode/train.py:372-439.  `batches`: list of dicts with "tag", "n" and label tensors.
    Returns (loss_sum, {name: (values, weights)}) - names in first-seen order."""
    vals, wts = defaultdict(list), defaultdict(list)
    offset, total = 0, 0
    for sub in batches:
        n = sub["n"]
        sp = {k: v[offset:offset + n] for k, v in preds.items()}
        for name, fn, w in criterions[sub["tag"]]:
            v = fn(sp, sub)
            wv = w(epoch) if callable(w) else w
            if "dataset_weight" in sub:
                wt = wv * sub["dataset_weight"]
            else:
                wt = v.new_full(v.shape, wv)
            vals[name].append(v)
            wts[name].append(wt)
        offset += n
        total += n
    by_name = {k: (torch.cat(vals[k]), torch.cat(wts[k])) for k in vals}
    loss_sum = torch.cat([v * w for v, w in by_name.values()]).sum() / total
    return loss_sum, by_name


# =============================================================================================
# optimiser + schedule  (scripts/train_poseestimator.py:114-167,442-445; train.py:611-629)
# =============================================================================================
def lr_factor(e: int, epochs: int) -> float:
    """ExponentialUpThenSteps(num_up=max(1,E//10), gamma=0.1, steps=[E//2]) as a LambdaLR factor."""
    num_up = max(1, epochs // 10)
    if e < num_up:
        return 1.0e-2 * math.exp(-math.log(1.0e-2) * (e + 1) / num_up)
    steps = [0, epochs // 2]
    return 0.1 ** [j for j, s in enumerate(steps) if e > s][-1]


class ClipAdam:
    """clip_grad_norm_(max_norm=1.0, 2-norm over ALL params) followed by torch.optim.Adam
    (betas .9/.999, eps 1e-8, no weight decay) with 2 non-empty groups (variance heads at 0.1 lr)."""

    def __init__(self, state: dict, lr=1.0e-3, epochs=200, max_norm=1.0):
        self.state, self.base_lr, self.epochs, self.max_norm = state, lr, epochs, max_norm
        vk = set(variance_param_keys(state))
        self.groups = [
            ([k for k in state if not is_buffer(k) and k not in vk], lr),
            (sorted(vk, key=list(state).index), 0.1 * lr),
        ]
        self.m = {k: torch.zeros_like(state[k]) for g, _ in self.groups for k in g}
        self.v = {k: torch.zeros_like(state[k]) for g, _ in self.groups for k in g}
        self.t = 0
        self.epoch = 0

    def lrs(self):
        f = lr_factor(self.epoch, self.epochs)
        return [lr * f for _, lr in self.groups]

    @torch.no_grad()
    def step(self):
        params = [self.state[k] for g, _ in self.groups for k in g if self.state[k].grad is not None]
        total = torch.linalg.vector_norm(
            torch.stack([torch.linalg.vector_norm(p.grad, 2.0) for p in params]), 2.0
        )
        coef = torch.clamp(self.max_norm / (total + 1.0e-6), max=1.0)
        self.t += 1
        b1, b2, eps = 0.9, 0.999, 1.0e-8
        bc1, bc2 = 1.0 - b1**self.t, 1.0 - b2**self.t
        for (keys, _), lr in zip(self.groups, self.lrs()):
            for k in keys:
                p = self.state[k]
                if p.grad is None:
                    continue
                g = p.grad * coef
                self.m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(eps)
                p.addcdiv_(self.m[k], denom, value=-lr / bc1)
        return total

    def end_epoch(self):
        self.epoch += 1

    def zero_grad(self):
        for g, _ in self.groups:
            for k in g:
                self.state[k].grad = None


def swa_update(avg: dict, new: dict, n_averaged: int) -> None:
    """train.py:447-467 -> torch.optim.swa_utils.AveragedModel(use_buffers=True): equal-weight running
    mean of parameters AND buffers: avg += (new - avg) / (n + 1); first call copies."""
    for k in avg:
        if n_averaged == 0:
            avg[k].copy_(new[k])
        elif avg[k].is_floating_point():
            avg[k].add_((new[k] - avg[k]) / (n_averaged + 1))
        else:  # integer buffers (num_batches_tracked): same formula in integer arithmetic
            avg[k].copy_(avg[k] + torch.div(new[k] - avg[k], n_averaged + 1, rounding_mode="trunc"))
