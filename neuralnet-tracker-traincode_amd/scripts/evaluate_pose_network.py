#!/usr/bin/env python
"""Pose / landmark error table of a trained network on a labelled face-crop set (reference:
scripts/evaluate_pose_network.py:205-291).  Data: an .npz in the layout of tests/golden/aflw2kmini.npz
(`oracle/tools/h5_to_npz.py` converts the reference's HDF5 files: image_bytes + image_lengths (encoded images),
rois [N,4], quats [N,4], coords [N,3], optional pt3d_68 [N,68,3]); h5py is not available in this image.

    python scripts/evaluate_pose_network.py model.ckpt --data aflw2k3d.npz [--device cuda]
"""
from __future__ import annotations

import argparse
import io
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from trackertraincode import eval as E  # noqa: E402
from trackertraincode.neuralnets import models  # noqa: E402


def iter_samples(path):
    from PIL import Image

    d = np.load(path)
    off = 0
    for i, n in enumerate(d["image_lengths"]):
        img = np.array(Image.open(io.BytesIO(d["image_bytes"][off:off + int(n)].tobytes())))
        off += int(n)
        s = {"image": torch.from_numpy(img), "roi": d["rois"][i].astype(np.float32), "pose": d["quats"][i].astype(np.float32),
             "coord": d["coords"][i].astype(np.float32)}
        if "pt3d_68" in d.files:
            s["pt3d_68"] = d["pt3d_68"][i].astype(np.float32)
        yield s


class _All:
    def __init__(self, metrics):
        self.metrics = metrics

    def update(self, preds, targets):
        for m in self.metrics.values():
            m.update(preds, targets)

    def compute(self):
        return {k: m.compute() for k, m in self.metrics.items()}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("filename")
    ap.add_argument("--data", required=True)
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--roi-expansion", type=float, default=1.1)
    args = ap.parse_args(argv)
    net = models.load_model(args.filename)
    pred = E.Predictor(net, focus_roi_expansion_factor=args.roi_expansion, device=args.device)
    metrics = {"euler": E.EulerAngleErrors(), "geodesic": E.GeodesicError(), "xys": E.NormalizedXYSError()}
    if net.enable_point_head:
        metrics["nme3d"] = E.UnweightedKptNME(3)
    res = pred.evaluate(_All(metrics), iter_samples(args.data))
    tab = E.pose_error_table(res["euler"], res["geodesic"])
    print(f"{'pitch':>8} {'yaw':>8} {'roll':>8} {'MAE':>8} {'geodesic':>9}   [deg]")
    print(f"{tab['pitch']:8.3f} {tab['yaw']:8.3f} {tab['roll']:8.3f} {tab['mae']:8.3f} {tab['geodesic']:9.3f}")
    xys = res["xys"].cpu().numpy().mean(0) * 100.0
    print(f"position error x/y/size: {xys[0]:.2f} / {xys[1]:.2f} / {xys[2]:.2f} % of the box width")
    if "nme3d" in res:
        print(f"landmark NME 3d: {res['nme3d'].cpu().numpy().mean() * 100.0:.2f} %")
    return tab


if __name__ == "__main__":
    main()
