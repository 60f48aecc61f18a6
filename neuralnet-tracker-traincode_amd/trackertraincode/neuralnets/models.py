"""The pose network (reference: neuralnets/models.py).

`NetworkWithPointHead` keeps the reference's constructor, `forward(x, coord_convention_id) -> dict`,
`get_config`, `name`, `input_resolution(s)`, `prepare_finetune`, `train` and - through the same
sub-module / parameter names - its state dict, so checkpoints interchange.
On CUDA tensors the forward is two autograd nodes: the HIP backbone (backbones/mobilenet_v1.py) and
ONE fused heads kernel (csrc/heads.hip) that replaces boxnet / posnet / quatnet / both local pose
offsets / the landmark head.  CPU tensors are accepted in eval mode only (export / inspection).
"""
from __future__ import annotations

from typing import Any, Dict, Tuple

import torch
import torch.nn as nn
from torch import Tensor

from . import _hipops
from . import io as _io
from . import negloglikelihood as NLL
from . import torchquaternion
from ..backbones.mobilenet_v1 import MobileNet
from .math import inv_smoothclip0, smoothclip0
from .modelcomponents import (
    DeformableHeadKeypoints,
    LocalToGlobalCoordinateOffset,
    freeze_norm_stats,
    rigid_transformation_25d,
)
from .rotrepr import Mat33Repr, QuatRepr


# ---------------------------------------------------------------------------------------------
# head modules: parameter containers + plain-torch forward (CPU eval/export path)
# ---------------------------------------------------------------------------------------------
class Landmarks3dOutput(nn.Module):
    """Reference :96-124."""

    def __init__(self, num_features, enable_uncertainty=False):
        super().__init__()
        self.enable_uncertainty = enable_uncertainty
        self.deformablekeypoints = DeformableHeadKeypoints(40, 10)
        self.shapenet = nn.Linear(num_features, self.deformablekeypoints.num_eigvecs)
        if enable_uncertainty:
            self.point_distrib_scales = NLL.DiagonalScaleParameter(68)
            self.shape_distrib_scales = NLL.DiagonalScaleParameter(50)

    def scales(self, pt3d_68: Tensor, shapeparam: Tensor) -> Dict[str, Tensor]:
        return {
            "pt3d_68_scales": self.point_distrib_scales()[None, :, None].expand_as(pt3d_68),
            "shapeparam_scales": self.shape_distrib_scales()[None, :].expand_as(shapeparam),
        }

    def forward(self, z, quats, coords) -> Dict[str, Tensor]:
        shapeparam = self.shapenet(z)
        pts = rigid_transformation_25d(quats, coords[..., :2], coords[..., 2:], self.deformablekeypoints(shapeparam))
        out = {"pt3d_68": pts, "shapeparam": shapeparam}
        if self.enable_uncertainty:
            out.update(self.scales(pts, shapeparam))
        return out


class DirectQuaternionWithNormalization(nn.Module):
    """Reference :127-150."""

    def __init__(self, num_features, enable_uncertainty=False):
        super().__init__()
        self.enable_uncertainty = enable_uncertainty
        self.linear = nn.Linear(num_features, 4, bias=True)
        self.linear.bias.data[torchquaternion.iw] = inv_smoothclip0(torch.as_tensor(0.1))
        if enable_uncertainty:
            self.uncertainty_net = NLL.FeaturesAsTriangularScale(num_features, 3)

    def forward(self, x) -> Dict[str, Tensor]:
        quats, unnormalized = QuatRepr.from_features(self.linear(x))
        out = {"unnormalized_quat": unnormalized, "rot": quats}
        if self.enable_uncertainty:
            out["pose_scales_tril"] = self.uncertainty_net(x)
        return out


class RotRepr6dWithNormalization(nn.Module):
    """Reference :153-174 (optional --enable-6drot head).  On CUDA tensors the arithmetic runs inside the fused heads
    kernel (csrc/heads.hip, enable_6drot); this module owns the parameters and the CPU/eval form."""

    def __init__(self, num_features, enable_uncertainty=False):
        super().__init__()
        self.enable_uncertainty = enable_uncertainty
        self.linear = nn.Linear(num_features, 6, bias=True)
        self.linear.bias.data[...] = 0.001 * torch.as_tensor([1.0, 0.0, 0.0, 0.0, 1.0, 0.0])
        if enable_uncertainty:
            self.uncertainty_net = NLL.FeaturesAsTriangularScale(num_features, 3)

    def forward(self, x) -> Dict[str, Tensor]:
        z = self.linear(x)
        out = {"unnormalized_6drepr": z, "rot": Mat33Repr.from_6drepr_features(z)}
        if self.enable_uncertainty:
            out["pose_scales_tril"] = self.uncertainty_net(x)
        return out


class BoundingBox(nn.Module):
    """Reference :177-197."""

    def __init__(self, num_features, enable_uncertainty=False):
        super().__init__()
        self.enable_uncertainty = enable_uncertainty
        self.linear = nn.Linear(num_features, 4)
        self.linear.bias.data[...] = torch.tensor([0.0, 0.0, 0.5, 0.5])
        if enable_uncertainty:
            self.scales = NLL.DiagonalScaleParameter(4)

    def forward(self, x: Tensor) -> Dict[str, Tensor]:
        z = self.linear(x)
        size = smoothclip0(z[..., 2:])
        out = {"roi": torch.cat((z[..., :2] - size, z[..., :2] + size), dim=-1)}
        if self.enable_uncertainty:
            out["roi_scales"] = self.scales()[None, :].expand_as(z)
        return out


class PositionSizeOutput(nn.Module):
    """Reference :200-215."""

    def __init__(self, num_features, enable_uncertainty=False):
        super().__init__()
        self.enable_uncertainty = enable_uncertainty
        self.linear_xy = nn.Linear(num_features, 2)
        self.linear_size = nn.Linear(num_features, 1)
        self.linear_size.bias.data.fill_(0.5)
        if enable_uncertainty:
            self.scales = NLL.FeaturesAsTriangularScale(num_features, 3)

    def forward(self, x: Tensor):
        out = {"coord": torch.cat((self.linear_xy(x), smoothclip0(self.linear_size(x))), dim=-1)}
        if self.enable_uncertainty:
            out["coord_scales"] = self.scales(x)
        return out


def create_pose_estimator_backbone(num_heads, config: str, args: Dict[str, Any]):
    """Reference :218-232."""
    if config == "mobilenetv1":
        return MobileNet(input_channel=1, num_classes=None, **args)
    if config == "resnet18":
        from ..backbones.resnet import resnet18

        return resnet18(**args)
    raise NotImplementedError(
        f"backbone {config!r}: only 'mobilenetv1' and 'resnet18' are in scope of the MI355X path "
        "(efficientnet_* / hybrid_vit are third-party model definitions outside BASELINE's configs)"
    )


class CnnNeck(nn.Module):
    """[B,F] -> num_heads aliases of the same tensor (reference :235-256; its Dropout is never applied)."""

    def __init__(self, num_heads, args: Dict[str, Any]):
        super().__init__()
        self.num_heads = num_heads
        self.dropout_prob = args.get("dropout_prob", 0.5)
        self.dropout = nn.Dropout(self.dropout_prob) if self.dropout_prob > 0.0 else nn.Identity()

    def forward(self, features: Tensor) -> tuple[Tensor, ...]:
        return features[:, None, :].expand(-1, self.num_heads, -1).unbind(dim=1)


class NetworkWithPointHead(nn.Module):
    NUM_DATASET_CONSTANTS = 8

    def __init__(self, enable_point_head=True, enable_face_detector=False, config="mobilenetv1", enable_uncertainty=False,
                 dropout_prob=None, use_local_pose_offset=True, backbone_args=None, enable_6drot=False):
        super().__init__()
        assert dropout_prob is None or dropout_prob in (0.0, 0.5)
        if enable_face_detector:
            raise NotImplementedError("enable_face_detector: unused by the training script (always False, :292)")
        self.enable_point_head = enable_point_head
        self.enable_face_detector = enable_face_detector
        self.finetune = False
        self.config = config
        self.enable_uncertainty = enable_uncertainty
        self.use_local_pose_offset = use_local_pose_offset
        self.enable_6drot = enable_6drot
        self._backbone_args = {} if backbone_args is None else backbone_args
        self._input_resolution = (129,)
        num_heads = 3 + (1 if enable_point_head else 0)

        self.convnet = create_pose_estimator_backbone(num_heads, config, self._backbone_args)
        F = self.convnet.num_features
        self.neck = CnnNeck(num_heads, self._backbone_args)
        self.boxnet = BoundingBox(F, enable_uncertainty)
        self.posnet = PositionSizeOutput(F, enable_uncertainty)
        self.quatnet = (RotRepr6dWithNormalization if enable_6drot else DirectQuaternionWithNormalization)(F, enable_uncertainty)
        self.local_pose_offset = LocalToGlobalCoordinateOffset(self.NUM_DATASET_CONSTANTS)
        self.local_pose_offset_kpts = LocalToGlobalCoordinateOffset(self.NUM_DATASET_CONSTANTS)
        if enable_point_head:
            self.landmarks = Landmarks3dOutput(F, enable_uncertainty)

    def get_config(self):
        return {
            "enable_point_head": self.enable_point_head,
            "enable_face_detector": self.enable_face_detector,
            "config": self.config,
            "enable_uncertainty": self.enable_uncertainty,
            "use_local_pose_offset": self.use_local_pose_offset,
            "backbone_args": self._backbone_args,
            "enable_6drot": self.enable_6drot,
        }

    @property
    def input_resolutions(self) -> Tuple[int, ...]:
        r = self._input_resolution
        return r if isinstance(r, tuple) else (r,)

    @property
    def input_resolution(self) -> int:
        r = self._input_resolution
        return r[0] if isinstance(r, tuple) else r

    @property
    def name(self) -> str:
        return type(self).__name__ + "_" + self.config

    # ---- the fused HIP heads -------------------------------------------------------------------
    def _linear_stack(self):
        """(weight, bias) pairs in the row order of csrc/head_math.h."""
        mods = [self.boxnet.linear, self.posnet.linear_xy, self.posnet.linear_size, self.quatnet.linear]
        if self.enable_uncertainty:
            mods += [self.posnet.scales.neck.lin, self.quatnet.uncertainty_net.neck.lin]
        if self.enable_point_head:
            mods += [self.landmarks.shapenet]
        flat = []
        for m in mods:
            flat += [m.weight, m.bias]
        return flat

    def _heads_hip(self, feat: Tensor, coord_convention_id: Tensor | None) -> Dict[str, Tensor]:
        unc, pt, off, r6 = self.enable_uncertainty, self.enable_point_head, self.use_local_pose_offset, self.enable_6drot
        kp = self.landmarks.deformablekeypoints.keypts if pt else None
        ke = self.landmarks.deformablekeypoints.keyeigvecs if pt else None
        outs = _hipops.HeadsFn.apply(feat, coord_convention_id, unc, pt, off, r6, kp, ke, self.local_pose_offset.p,
                                     self.local_pose_offset_kpts.p if pt else None, *self._linear_stack())
        roi, coord, rot, qu = outs[:4]
        # key order of the reference's dict (:345-372; it is ExportModel's output order): with the local pose offset "rot" and "coord" are
        # popped and re-inserted behind the heads' other keys, without it they stay where posnet / quatnet put them
        out: Dict[str, Tensor] = {"roi": roi}
        k = 4
        if unc:
            out["roi_scales"] = self.boxnet.scales()[None, :].expand_as(roi)
        if not off:
            out["coord"] = coord
        if unc:
            out["coord_scales"] = outs[k]
            k += 2
        out["unnormalized_6drepr" if r6 else "unnormalized_quat"] = qu
        if not off:
            out["rot"] = Mat33Repr(rot) if r6 else QuatRepr(rot)
        if unc:
            out["pose_scales_tril"] = outs[5]
        if off:
            out["rot"] = Mat33Repr(rot) if r6 else QuatRepr(rot)
            out["coord"] = coord
        if pt:
            out["pt3d_68"], out["shapeparam"] = outs[k], outs[k + 1]
            if unc:
                out.update(self.landmarks.scales(outs[k], outs[k + 1]))
        return out

    def _heads_torch(self, x: Tensor, coord_convention_id: Tensor | None) -> Dict[str, Tensor]:
        """Reference :345-372 in plain torch ops (CPU eval/export)."""
        zs = list(self.neck(x))
        out: Dict[str, Tensor] = self.boxnet(zs.pop())
        out.update(self.posnet(zs.pop()))
        out.update(self.quatnet(zs.pop()))
        hidden_rot, hidden_coord = out["rot"], out["coord"]
        if self.use_local_pose_offset:
            # (pop + re-insert as the reference does, :352-356: "rot" and "coord" move behind the heads' other keys - the key ORDER of
            # the eval-mode dict is the output order of scripts/export_model.py's ExportModel)
            out["rot"], out["coord"] = self.local_pose_offset(out.pop("rot"), out.pop("coord"), set_id=coord_convention_id)
        if self.enable_point_head:
            rots, coords = out["rot"], out["coord"]
            if self.use_local_pose_offset:
                rots, coords = self.local_pose_offset_kpts(hidden_rot, hidden_coord, set_id=coord_convention_id)
            out.update(self.landmarks(zs.pop(), rots, coords))
        return out

    def forward(self, x: Tensor, coord_convention_id: Tensor | None = None):
        assert x.shape[2] in self.input_resolutions and x.shape[3] == x.shape[2]
        if x.is_cuda:
            feat = self.convnet.forward_features(x) if hasattr(self.convnet, "forward_features") else self.convnet(x)[0]
            out = self._heads_hip(feat, coord_convention_id)
        else:
            if self.training:
                raise RuntimeError("the MI355X training path needs CUDA tensors; CPU tensors are accepted in eval() mode only")
            feat, _ = self.convnet(x)
            out = self._heads_torch(feat, coord_convention_id)
        if not self.training:
            out["pose"] = out["rot"].as_quat()
        return out

    def prepare_finetune(self):
        """Parameter groups for fine-tuning with per-layer learning rates (reference :378-389)."""
        self.finetune = True
        groups = [list(c.parameters()) for c in sum((list(c.children()) for c in self.convnet.children()), [])]
        taken = frozenset(sum(groups, []))
        groups.append([p for p in self.parameters() if p not in taken])
        return groups

    def train(self, mode=True):
        super().train(mode)
        if mode and self.finetune:
            # reference :391-394: the backbone's normalisation layers stay in eval mode with frozen affine parameters; the HIP
            # backbone then runs its backward through the fixed affine maps (ttk_bn_bwd_frozen)
            self.convnet.apply(freeze_norm_stats)
        return self


save_model = _io.save_model


def load_model(filename: str):
    """Reference :399-415 (the legacy bare-state-dict format is loaded as the reference's legacy config)."""
    try:
        return _io.load_model(filename, [NetworkWithPointHead])
    except _io.InvalidFileFormatError as e:
        print(f"Failed to load model because: {str(e)}. Will attempt to load legacy config")
        net = NetworkWithPointHead(enable_point_head=True, enable_face_detector=False, config="resnet18", enable_uncertainty=True,
                                   backbone_args={"use_blurpool": False})
        net.load_state_dict(torch.load(filename), strict=True)
        return net
