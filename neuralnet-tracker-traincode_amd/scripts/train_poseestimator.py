#!/usr/bin/env python
"""Training entry point with the reference's flags (reference: scripts/train_poseestimator.py), run on
MI355X GPUs through the HIP kernels.  `python train_poseestimator.py --ds synthetic --epochs 2`.

Same functions as the reference script for the parts the hot path needs: `setup_losses` (:170-285),
`find_variance_parameters` / `setup_lr_with_slower_variance_training` / `create_optimizer` (:114-167),
`create_net` (:288-296).  Lightning's Trainer is replaced by trackertraincode.train.fit; plotting and
the HDF5 datasets are not part of this package.

Data-parallel replicas (no counterpart in the reference, which trains on one device): launch one process per GPU with
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/train_poseestimator.py ...
Every rank then trains on its own stream of batches (--batchsize is per GPU), gradients are all-reduced over RCCL while
backward runs (trackertraincode.parallel), and rank 0 writes the checkpoints.
"""
from __future__ import annotations

import argparse
import os
import sys
from os.path import dirname, join

import torch
import torch.nn as nn

sys.path.insert(0, dirname(dirname(os.path.abspath(__file__))))

import trackertraincode.neuralnets.losses as losses  # noqa: E402
import trackertraincode.neuralnets.models as models  # noqa: E402
import trackertraincode.neuralnets.negloglikelihood as NLL  # noqa: E402
import trackertraincode.pipelines  # noqa: E402
import trackertraincode.train as train  # noqa: E402
from trackertraincode.pipelines import Tag  # noqa: E402


def find_variance_parameters(net: nn.Module):
    """Parameters of the uncertainty heads: trained at 0.1 x lr (reference :114-122)."""
    if isinstance(net, (NLL.FeaturesAsTriangularScale, NLL.FeaturesAsDiagonalScale, NLL.DiagonalScaleParameter)):
        return list(net.parameters())
    return sum((find_variance_parameters(c) for c in net.children()), start=[])


def find_transformer_parameters(net: nn.Module):
    if isinstance(net, (nn.TransformerEncoderLayer, nn.TransformerDecoderLayer)):
        return list(net.parameters())
    return sum((find_transformer_parameters(c) for c in net.children()), start=[])


def setup_lr_with_slower_variance_training(net, base_lr):
    variance = find_variance_parameters(net)
    transformer = find_transformer_parameters(net)
    special = {id(p) for p in variance + transformer}
    other = [p for p in net.parameters() if id(p) not in special]
    return [
        {"params": other, "lr": base_lr},
        {"params": variance, "lr": 0.1 * base_lr},
        {"params": transformer, "lr": 0.01 * base_lr, "weight_decay": 0.01},
    ]


def create_optimizer(net, args):
    """Fused clip(1.0)+Adam over the reference's three parameter groups and its LR schedule."""
    groups = [g for g in setup_lr_with_slower_variance_training(net, args.lr) if g["params"]]
    optimizer = train.ClipAdam(groups, lr=args.lr, max_norm=1.0)
    n_epochs = args.epochs
    scheduler = train.ExponentialUpThenSteps(optimizer, max(1, n_epochs // 10), 0.1, [n_epochs // 2])
    return optimizer, scheduler


def setup_losses(args, net):
    """Tag -> CriterionGroup tables with the reference's names and weights (:170-285)."""
    C = train.Criterion
    rot_loss = losses.Rot6dReprLoss() if args.enable_6drot else losses.QuatPoseLoss("approx_distance")
    rot_constraint = losses.Rot6dNormalizationSoftConstraint() if args.enable_6drot else losses.QuaternionNormalizationSoftConstraint()
    cregularize = [C("quatregularization1", rot_constraint, 1.0e-6)]
    poselosses, roilosses, pointlosses, pointlosses25d, shapeparamloss = [], [], [], [], []

    if args.with_nll_loss:
        def ramped(multiplier):
            if not args.rampup_nll_losses:
                return multiplier * 0.01
            return lambda step: 0.01 * min(1.0, max(0.0, (step / args.epochs - 0.1) * 10.0)) * multiplier

        poselosses += [C("nllrot", NLL.QuatPoseNLLLoss(), ramped(0.5)), C("nllcoord", NLL.CorrelatedCoordPoseNLLLoss(), ramped(0.5))]
        if args.with_roi_train:
            roilosses += [C("nllbox", NLL.BoxNLLLoss(distribution="gaussian"), ramped(0.01))]
        if args.with_pointhead:
            pointlosses += [C("nllpoints3d", NLL.Points3dNLLLoss(chin_weight=0.8, eye_weight=0.0), ramped(0.5))]
            pointlosses25d += [C("nllpoints3d", NLL.Points3dNLLLoss(chin_weight=0.8, eye_weight=0.0, pointdimension=2), ramped(0.5))]
    poselosses += [C("rot", rot_loss, 1.0), C("xy", losses.PoseXYLoss("l2"), 0.25), C("sz", losses.PoseSizeLoss("l2"), 0.25)]
    if args.with_roi_train:
        roilosses += [C("box", losses.BoxLoss("l2"), 0.01)]
    if args.with_pointhead:
        pointlosses += [C("points3d", losses.Points3dLoss("l2", chin_weight=0.8, eye_weights=0.0), 0.5)]
        pointlosses25d += [C("points3d", losses.Points3dLoss("l2", pointdimension=2, chin_weight=0.8, eye_weights=0.0), 0.5)]
        shapeparamloss += [C("shp_l2", losses.ShapeParameterLoss(), 0.1)]
        cregularize += [C("nll_shp_gmm", losses.ShapePlausibilityLoss(), 0.1)]

    G = train.CriterionGroup
    train_criterions = {
        Tag.ONLY_POSE: G(poselosses + cregularize + roilosses),
        Tag.POSE_WITH_LMKS_NO_SHAPE_PARAMS: G(poselosses + cregularize + pointlosses + roilosses),
        Tag.POSE_WITH_LANDMARKS: G(poselosses + cregularize + pointlosses + shapeparamloss + roilosses),
        Tag.POSE_WITH_LANDMARKS_3D_AND_2D: G(poselosses + cregularize + pointlosses + shapeparamloss + roilosses),
        Tag.ONLY_LANDMARKS: G(pointlosses + cregularize),
        Tag.ONLY_LANDMARKS_25D: G(pointlosses25d + cregularize),
    }
    test_criterions = {Tag.POSE_WITH_LANDMARKS: G(poselosses + pointlosses + roilosses + shapeparamloss + cregularize)}
    return train_criterions, test_criterions


def create_net(args):
    return models.NetworkWithPointHead(
        enable_point_head=args.with_pointhead, enable_face_detector=False, config=args.backbone,
        enable_uncertainty=args.with_nll_loss, backbone_args={"use_blurpool": args.with_blurpool}, enable_6drot=args.enable_6drot,
    )


def parse_dataset_definition(arg: str):
    """CLI dataset specification <name1>[:<weight1>]+<name2>[:<weight2>]+... -> (ids in first-seen order, {id: weight}) with the reference's
    names (:60-94; the reference returns the ids as list(frozenset(...)), i.e. in hash order - the loaders sort them by their own table)."""
    Id = trackertraincode.pipelines.Id
    dsmap = {"300wlp": Id._300WLP, "synface": Id.SYNFACE, "aflw2k": Id.AFLW2k3d, "biwi": Id.BIWI, "wider": Id.WIDER, "repro_300_wlp": Id.REPO_300WLP,
             "repro_300_wlp_woextra": Id.REPO_300WLP_WO_EXTRA, "wflw_lp": Id.WFLW_LP, "lapa_megaface_lp": Id.LAPA_MEGAFACE_LP,
             "panoptic": Id.PANOPTIC_CMU, "replicantface": Id.REPLICANT_FACE}
    splitted = arg.split("+")
    unknown = [s.split(":")[0] for s in splitted if s.split(":")[0] not in dsmap]
    if unknown:
        raise ValueError(f"unknown dataset(s) {unknown}; available: synthetic, {', '.join(dsmap)}")
    dataset_weights = {dsmap[k]: float(v) for k, v in (tuple(s.split(":")) for s in splitted if ":" in s)}
    dsids = list(dict.fromkeys(dsmap[s.split(":")[0]] for s in splitted))
    return dsids, dataset_weights


def setup_datasets(args, device, rank=0):
    """(train_loader, test_loader, size) - reference :66-78.  "synthetic": seeded synthetic crops; otherwise dataset ids read from the
    converted shards under $DATADIR (trackertraincode.pipelines.make_pose_estimation_loaders)."""
    common = dict(inputsize=args.input_size, batchsize=args.batchsize, device=device, seed=1234 + rank, enable_image_aug=args.with_image_aug,
                  rotation_aug_angle=args.rotation_aug_angle, roi_override=args.roi_override)
    if args.ds == "synthetic":
        return trackertraincode.pipelines.make_pose_estimation_loaders(datasets="synthetic", **common)
    ids, weights = parse_dataset_definition(args.ds)
    return trackertraincode.pipelines.make_pose_estimation_loaders(datasets=ids, dataset_weights=weights,
                                                                   use_weights_as_sampling_frequency=args.ds_weight_are_sampling_frequencies, **common)


def make_parser():
    p = argparse.ArgumentParser(description="Trains the model")
    p.add_argument("--backbone", default="mobilenetv1")
    p.add_argument("--batchsize", type=int, default=64)
    p.add_argument("--lr", type=float, default=1.0e-3)
    p.add_argument("--epochs", type=int, default=200)
    p.add_argument("--ds", type=str, default="synthetic")
    p.add_argument("--with-swa", action="store_true", default=False, dest="swa")
    p.add_argument("--outdir", type=str, default=join(dirname(__file__), "..", "model_files"))
    p.add_argument("--ds-weighting", action="store_false", default=True, dest="ds_weight_are_sampling_frequencies")
    p.add_argument("--no-pointhead", action="store_false", default=True, dest="with_pointhead")
    p.add_argument("--with-nll-loss", default=False, action="store_true")
    p.add_argument("--raug", default=30, type=float, dest="rotation_aug_angle")
    p.add_argument("--no-imgaug", default=True, action="store_false", dest="with_image_aug")
    p.add_argument("--blurpool", default=False, action="store_true", dest="with_blurpool")
    p.add_argument("--roi-override", default="original", type=str, choices=["extent_to_forehead", "original", "landmarks"], dest="roi_override")
    p.add_argument("--no-roi-train", default=True, action="store_false", dest="with_roi_train")
    p.add_argument("--rampup-nll-losses", default=False, action="store_true")
    p.add_argument("--enable-6drot", default=False, action="store_true")
    # not a flag of the reference (it trains in fp32 only): storage of the backbone's activations in HBM
    p.add_argument("--precision", default="fp32",
                   help="fp32 (the reference's precision) | bf16-compute (mobilenetv1): activations and their gradients bfloat16 in 64-channel blocks AND bf16 "
                   "operands of the pointwise convolutions (one MFMA product, fp32 accumulation; master weights, statistics and Adam fp32).  The "
                   "storage-only variants bf16 / bf16-all of earlier rounds are retired (they were slower than fp32): they raise, naming bf16-compute")
    p.add_argument("--graph-steps", default=False, action="store_true",
                   help="single GPU: replay one captured hipGraph per training step instead of ~150 eager launches (train.GraphedTrainStep)")
    return p


def main():
    import torch.distributed as dist
    from trackertraincode import parallel

    args = make_parser().parse_args()
    args.input_size = 129
    from trackertraincode.backbones import mobilenet_v1

    mobilenet_v1.set_activation_dtype(args.precision)
    world, rank = int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("RANK", 0))
    # TTK_DRYRUN_SHARE_GPU=1: every rank on device 0 with gloo instead of RCCL - a dry run of the multi-rank wiring on a one-GPU box
    # (tests/test_train_script_dp_dryrun_gpu.py); RCCL needs one device per rank
    share_gpu = os.environ.get("TTK_DRYRUN_SHARE_GPU", "0") != "0"
    device = torch.device("cuda", 0 if share_gpu else int(os.environ.get("LOCAL_RANK", 0)))
    torch.cuda.set_device(device)
    if world > 1:  # before anything else touches the GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    train_loader, test_loader, _ = setup_datasets(args, device, rank)
    net = create_net(args).to(device)
    parallel.broadcast_module_state(net)  # every replica starts from rank 0's weights
    train_crit, test_crit = setup_losses(args, net)
    optimizer, scheduler = create_optimizer(net, args)
    out_dir = join(args.outdir, net.name)
    # ModelCheckpoint(monitor="val_loss", filename="best", save_last=True) of the reference (:423-431): rank 0 writes best.ckpt / last.ckpt
    callbacks = [train.CheckpointCallback(out_dir)] if rank == 0 else []
    if args.swa and rank == 0:
        callbacks.append(train.SwaCallback(start_epoch=args.epochs * 2 // 3))
    reducer = None
    if world > 1:
        reducer = parallel.GradAllReduce()
        parallel.install(reducer)                  # gradient arenas are all-reduced in place while backward runs
        optimizer.grad_scale = reducer.grad_scale  # 1/world inside the fused clip+Adam kernel

    try:
        train.fit(net, train_loader, train_crit, optimizer, scheduler, epochs=args.epochs, callbacks=callbacks,
                  val_loader=test_loader if rank == 0 else None, val_criterions=test_crit, reducer=reducer,
                  graphed=bool(getattr(args, "graph_steps", False)) and reducer is None)
    finally:
        parallel.install(None)
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        if not os.path.exists(join(out_dir, "last.ckpt")):  # (no validation epoch ran)
            models.save_model(net.to("cpu"), join(out_dir, "last.ckpt"))
        for cb in callbacks:
            if hasattr(cb, "on_train_end"):
                cb.on_train_end(out_dir)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
