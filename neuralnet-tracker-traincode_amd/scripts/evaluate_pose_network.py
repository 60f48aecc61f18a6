#!/usr/bin/env python
"""Pose / landmark error tables of trained networks on the reference's validation sets (reference: scripts/evaluate_pose_network.py):
one row per (model, data set, box configuration) with mean absolute pitch / yaw / roll and their mean, the geodesic error (degrees),
the RMSE of position and size in percent of the box width, the 3D landmark NME and the 2D NME binned by |yaw| (0-30-60-90 degrees).

    python scripts/evaluate_pose_network.py run1/best.ckpt run2/best.ckpt --ds aflw2k3d+biwi [--comprehensive-roi] [--json table.json]
           [--alignment-scheme none|perspective|opal23] [--roi-expansion 1.1] [--device cuda] [--datadir DIR]

The crops, the network and the back-transformation run on the MI355X (`trackertraincode.eval.Predictor`).  Data: `<datadir>/<name>.npz`
shards converted once from the reference's HDF5 files (oracle/tools/h5_to_npz.py; h5py is not in this image), default $DATADIR; a path to
an .npz given as --ds is read as is (layout of tests/golden/aflw2kmini.npz).  Box configurations as in the reference: "(H_roi)" = extent of
the posed BFM head mesh (needs the BFM blob, trackertraincode/facemodel/bfm.py; the reference's default), "(F_roi)" = extent of the 68
landmarks; ROI<f> = enlargement of the crop.  Without the blob the default falls back to "(F_roi)" with a note on stderr - a substitution
that is visible in the row's name.  Not built: `--vis` (the reference's matplotlib browser of the worst samples) and ONNX model files
(onnxruntime is not in this image): checkpoints only."""
from __future__ import annotations

import argparse
import json
import os
import sys
from collections import defaultdict
from os.path import commonprefix, relpath
from typing import NamedTuple

import numpy as np
import tabulate
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from trackertraincode import eval as E  # noqa: E402
from trackertraincode import pipelines, utils  # noqa: E402
from trackertraincode.neuralnets import models  # noqa: E402

BIWI_HORIZONTAL_FOV = 57.0  # the Kinect's horizontal field of view in degrees (reference :40-42)


class RoiConfig(NamedTuple):
    expansion_factor: float = 1.1
    center_crop: bool = False
    use_head_roi: bool = True

    def __str__(self):
        return f'{"(H_roi)" if self.use_head_roi else "(F_roi)"}{"CC" if self.center_crop else "ROI"}{self.expansion_factor:0.1f}'


comprehensive_roi_configs = [RoiConfig(f, False, head) for head in (True, False) for f in (1.2, 1.1, 1.0)]


class TableBuilder:
    """Rows per model; `build()` renders one github-style table per model, `build_json()` the columns per model (reference :107-193)."""

    data_name_table = {"aflw2k3d": "AFLW 2k 3d", "aflw2k3d_grimaces": "grimaces"}
    header = ["Data", "Pitch°", "Yaw°", "Roll°", "Mean°", "Geodesic°", "XY%", "S%", "NME3d%", "NME2d%_30", "NME2d%_60", "NME2d%_90", "NME2d%_avg"]

    def __init__(self):
        self._entries_by_model = defaultdict(list)

    def add_row(self, model: str, data: str, euler_angles, geodesic, rmse_pos, rmse_size, unweighted_nme_3d, nme_2d, data_aux_string=None):
        # (the reference's stand-ins for missing landmark columns: "n/a" and, for the 2D bins, "/na")
        nme3d = float(unweighted_nme_3d) * 100 if unweighted_nme_3d is not None else "n/a"
        nme2d = ["/na"] * 4 if nme_2d is None else [float(x) * 100 for x in nme_2d]
        name = self.data_name_table.get(data, data) + (data_aux_string if data_aux_string is not None else "")
        euler_angles = [float(a) for a in euler_angles]
        self._entries_by_model[model].append([name] + euler_angles + [float(np.average(euler_angles)), float(geodesic), float(rmse_pos), float(rmse_size),
                                                                      nme3d] + nme2d)

    def build(self) -> str:
        prefix = commonprefix(list(self._entries_by_model.keys()))
        lines = []
        for model, rows in self._entries_by_model.items():
            lines.append(relpath(model, prefix))
            lines += tabulate.tabulate(rows, self.header, tablefmt="github", floatfmt=".2f").splitlines()
        return "\n".join(lines)

    def build_json(self) -> str:
        """(Plain Python floats throughout: the reference hands numpy float32 scalars to json.dumps for the landmark columns and raises.)"""
        prefix = commonprefix([os.path.dirname(m) for m in self._entries_by_model])
        table = {}
        for model, rows in self._entries_by_model.items():
            cols = defaultdict(list)
            for row in rows:
                for name, value in zip(self.header, row):
                    cols[name].append(value)
            table[relpath(model, prefix)] = cols
        return json.dumps(table, indent=2)


class _Metrics:
    """The dict-of-metrics role of torchmetrics.MetricCollection."""

    def __init__(self, metrics):
        self.metrics = dict(metrics)

    def update(self, preds, targets):
        for m in self.metrics.values():
            m.update(preds, targets)

    def compute(self):
        return {k: m.compute() for k, m in self.metrics.items()}


def _npz_samples(path):
    """A bare .npz of labelled frames (tests/golden/aflw2kmini.npz): stored boxes, no filtering."""
    from trackertraincode.datasets.shards import decode_pose_shard

    return pipelines.ValidationSamples(decode_pose_shard(path), np.arange(len(np.load(path)["rois"])), lambda s: s)


_NETS: dict = {}


def load_pose_network(filename, device):
    if filename.endswith(".onnx"):
        raise NotImplementedError("ONNX model files need onnxruntime, which this image does not have: evaluate the checkpoint")
    if (filename, device) not in _NETS:
        _NETS.clear()  # one network at a time, like the reference's lru_cache(maxsize=1)
        _NETS[(filename, device)] = models.load_model(filename).to(device).eval()
    return _NETS[(filename, device)]


def evaluate(net_filename, data_name, roi_config: RoiConfig, args) -> dict:
    """The metric outputs of one (model, data, box configuration): what the reference's `predictor.evaluate(metrics, loader)` returns."""
    if roi_config.center_crop:
        raise NotImplementedError("center-crop configurations: the reference defines the flag and never evaluates with it (:69-73 is unused)")
    if data_name.endswith(".npz"):
        samples = _npz_samples(data_name)
    else:
        samples = pipelines.make_validation_loader(data_name, use_head_roi=roi_config.use_head_roi, return_single_samples=True, datadir=args.datadir)
    net = load_pose_network(net_filename, args.device)
    predictor = E.Predictor(net, roi_config.expansion_factor, device=args.device)
    metrics = {"pose_errs": E.NormalizedXYSError()}
    if args.alignment_scheme == "none":
        metrics.update(geodesic_errs=E.GeodesicError(), euler_errs=E.EulerAngleErrors())
    else:
        metrics.update(geodesic_errs=E.AlignedRotationErrorMetric("geo", args.alignment_scheme, BIWI_HORIZONTAL_FOV),
                       euler_errs=E.AlignedRotationErrorMetric("euler", args.alignment_scheme, BIWI_HORIZONTAL_FOV))
    first = next(iter(samples))
    if "pt3d_68" in first and getattr(net, "enable_point_head", False):  # model and data both have landmarks (reference :196-204)
        metrics.update(uw_nme_3d=E.UnweightedKptNME(), nme_2d=E.KptNME(dimensions=2))
    return predictor.evaluate(_Metrics(metrics), samples)


def report(net_filename, data_name, roi_config: RoiConfig, args, builder: TableBuilder):
    """One table row (reference :205-256)."""
    results = evaluate(net_filename, data_name, roi_config, args)
    pose_errs = np.asarray(torch.as_tensor(results["pose_errs"]).cpu())
    geodesic = np.asarray(torch.as_tensor(results["geodesic_errs"]).cpu())
    euler = np.asarray(torch.as_tensor(results["euler_errs"]).cpu())
    uw_nme_3d = np.asarray(torch.as_tensor(results["uw_nme_3d"]).cpu()) if "uw_nme_3d" in results else None
    e_x, e_y, e_size = pose_errs.T
    builder.add_row(
        model=net_filename, data=data_name,
        euler_angles=(np.average(np.abs(euler), axis=0) * utils.rad2deg).tolist(),
        geodesic=np.average(geodesic) * utils.rad2deg,
        rmse_pos=np.sqrt(np.average(np.square(e_x) + np.square(e_y))) * 100.0,
        rmse_size=np.sqrt(np.average(np.square(e_size))) * 100.0,
        unweighted_nme_3d=np.average(uw_nme_3d) if uw_nme_3d is not None else None,
        nme_2d=results.get("nme_2d"),
        # (a bare .npz is evaluated with the boxes it stores: neither the head-mesh nor the landmark extent is substituted)
        data_aux_string=" / " + (f"(stored)ROI{roi_config.expansion_factor:0.1f}" if data_name.endswith(".npz") else str(roi_config)),
    )


def _have_bfm_blob():
    from trackertraincode.facemodel import bfm

    return os.path.exists(os.path.join(bfm._FOLDER, "bfm_noneck_v3.pkl"))


def run(args) -> TableBuilder:
    builder = TableBuilder()
    if args.comprehensive_roi:
        assert args.roi_expansion is None, "Conflicting arguments"
        roi_configs = list(comprehensive_roi_configs)
    else:
        roi_configs = [RoiConfig(expansion_factor=args.roi_expansion) if args.roi_expansion is not None else RoiConfig()]
    if any(c.use_head_roi for c in roi_configs) and not _have_bfm_blob() and not all(d.endswith(".npz") for d in args.ds.split("+")):
        if not getattr(args, "allow_landmark_roi_fallback", False):
            raise FileNotFoundError("the (H_roi) box configurations need the BFM head mesh (trackertraincode/facemodel/bfm_noneck_v3.pkl, which the "
                                    "reference's repository does not carry either); pass --allow-landmark-roi-fallback to evaluate them as (F_roi) - "
                                    "the substitution shows in the row names, also of the JSON output")
        print("note: no BFM head mesh (trackertraincode/facemodel/bfm_noneck_v3.pkl): the (H_roi) configurations are evaluated as (F_roi)", file=sys.stderr)
        seen, repl = set(), []
        for c in roi_configs:
            c = c._replace(use_head_roi=False)
            if c not in seen:
                seen.add(c)
                repl.append(c)
        roi_configs = repl
    for net_filename in args.filenames:
        for name in args.ds.split("+"):
            for roi_config in roi_configs:
                report(net_filename, name, roi_config, args, builder)
    if args.json:
        assert args.json.endswith(".json")
        print(f"writing {args.json}")
        with open(args.json, "w") as f:
            f.write(builder.build_json())
    else:
        print(builder.build())
    return builder


def make_parser():
    ap = argparse.ArgumentParser(description="Evaluate pose networks")
    ap.add_argument("filenames", help="checkpoint files", type=str, nargs="*")
    ap.add_argument("--vis", default="none", choices=["none", "kpts", "rot", "size"])
    ap.add_argument("--device", help="cuda (the MI355X) - the crop and the network run in HIP kernels", default="cuda", type=str)
    ap.add_argument("--comprehensive-roi", action="store_true", default=False)
    ap.add_argument("--alignment-scheme", choices=["perspective", "opal23", "none"], default="none")
    ap.add_argument("--roi-expansion", default=None, type=float)
    ap.add_argument("--json", type=str, default=None)
    ap.add_argument("--allow-landmark-roi-fallback", action="store_true", default=False,
                    help="without the BFM head-mesh blob: evaluate the (H_roi) configurations with the landmark extent (F_roi) instead of failing")
    ap.add_argument("--ds", type=str, default="aflw2k3d", help="validation sets joined by '+', or paths of .npz files")
    ap.add_argument("--datadir", type=str, default=None, help="directory of the converted shards (default $DATADIR)")
    return ap


def main(argv=None):
    args = make_parser().parse_args(argv)
    if args.vis != "none":
        raise NotImplementedError("--vis: the reference's matplotlib browser of the worst samples is not part of this package")
    return run(args)


if __name__ == "__main__":
    main()
