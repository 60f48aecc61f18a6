"""CPU checks of the data / harness pieces around the training path: shard decoding, the validation epoch's
reduction and the checkpoint callback (reference scripts/train_poseestimator.py:332-338, 423-431).  The end-to-end run on the GPU is
tests/test_fit_shards_gpu.py."""
import os

import numpy as np
import pytest
import torch

from util import GOLDEN


def test_shard_decoding_matches_the_hdf5_contents():
    from trackertraincode.datasets.shards import decode_pose_shard

    raw = np.load(os.path.join(GOLDEN, "aflw2kmini.npz"))
    s = decode_pose_shard(os.path.join(GOLDEN, "aflw2kmini.npz"))
    assert s["image"].dtype == np.uint8 and s["image"].shape[:2] == (16, 1) and s["image"].shape[2:] == tuple(s["image_size"].max(0)[::-1])
    assert s["image"].std() > 20  # decoded pictures, not zeros
    np.testing.assert_array_equal(s["roi"], raw["rois"])
    np.testing.assert_array_equal(s["pose"], raw["quats"])
    np.testing.assert_array_equal(s["shapeparam"], raw["shapeparams"])
    np.testing.assert_allclose(s["coord"], raw["coords"] + np.array([0.5, 0.5, 0.0], np.float32))  # cell-centred pixels (normalization.py:83-90)
    np.testing.assert_allclose(s["pt3d_68"], raw["pt3d_68"] + np.array([0.5, 0.5, 0.0], np.float32))



def test_validate_reduction_and_checkpoint_callback(tmp_path):
    """validate(): per batch SUM over samples and terms of value * weight, per epoch the batch-size-weighted mean (Lightning's
    on_epoch reduction of self.log(..., batch_size=n)); the criterions get the BATCH INDEX as step; eval mode during, train mode after.
    CheckpointCallback: last.ckpt every validation epoch, best.ckpt at every new minimum."""
    import trackertraincode.train as train
    from trackertraincode.datasets.batch import Batch, Metadata
    from trackertraincode.neuralnets.models import NetworkWithPointHead, load_model
    from trackertraincode.pipelines import Tag

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.tensor(2.0))
            self.modes = []

        def forward(self, x):
            self.modes.append(self.training)
            return {"v": x.flatten(1).sum(1) * self.w}

    steps = []
    crit = {Tag.ONLY_POSE: train.CriterionGroup([train.Criterion("a", lambda p, b: p["v"], 0.5),
                                                 train.Criterion("b", lambda p, b: p["v"] ** 2, lambda step: steps.append(step) or float(step + 1))])}
    batches = [Batch(Metadata(4, 3, Tag.ONLY_POSE), image=torch.arange(12.0).reshape(3, 1, 2, 2)),
               Batch(Metadata(4, 1, Tag.ONLY_POSE), image=torch.ones(1, 1, 2, 2))]
    net = Net().train()
    got = train.validate(net, batches, crit)
    v0, v1 = batches[0]["image"].flatten(1).sum(1) * 2.0, batches[1]["image"].flatten(1).sum(1) * 2.0
    want = (float((0.5 * v0).sum() + (1.0 * v0 ** 2).sum()) * 3 + float((0.5 * v1).sum() + (2.0 * v1 ** 2).sum()) * 1) / 4
    assert abs(got - want) <= 1e-6 * want
    assert steps == [0, 1] and net.modes == [False, False] and net.training

    real = NetworkWithPointHead(enable_point_head=False, config="mobilenetv1", backbone_args={"use_blurpool": False})
    ck = train.CheckpointCallback(str(tmp_path / "out"))
    for epoch, v in enumerate([3.0, 2.0, 2.5]):
        with torch.no_grad():
            real.boxnet.linear.bias.fill_(float(epoch))
        ck.on_validation_end(epoch, real, v)
    assert ck.best_epoch == 1 and ck.best_value == 2.0 and ck.history == [3.0, 2.0, 2.5]
    best, last = load_model(ck.best_model_path), load_model(ck.last_model_path)
    assert float(best.boxnet.linear.bias[0]) == 1.0 and float(last.boxnet.linear.bias[0]) == 2.0
    assert best.get_config() == real.get_config()
