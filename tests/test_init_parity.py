"""Seed parity of freshly constructed models: under torch.manual_seed(s) the build's
NetworkWithPointHead must hold exactly the values the reference's holds - the custom conv init law
N(0, sqrt(2/(kh*kw*Cout))) for EVERY Conv2d (reference backbones/mobilenet_v1.py:155-158), the head bias overrides
(models.py:132,159,182,206) and the order in which the modules draw from torch's generator.  The fixture
tests/golden/init.npz was produced by oracle/tools/gen_golden.py from the imported reference."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.synth import digest


@pytest.mark.parametrize("seed", [0, 7])
def test_fresh_model_equals_reference_under_seed(seed, golden_dir):
    from trackertraincode.neuralnets.models import NetworkWithPointHead

    d = np.load(os.path.join(golden_dir, "init.npz"))
    cfg = json.loads(str(d["meta"]))["config"]
    torch.manual_seed(seed)
    net = NetworkWithPointHead(**cfg)
    sd = net.state_dict()
    keys = [k[len(f"seed{seed}/"):] for k in d.files if k.startswith(f"seed{seed}/")]
    assert set(keys) == {k for k in sd if not (k.endswith("keypts") or k.endswith("keyeigvecs"))}
    for k in keys:
        mine = digest(sd[k].detach().numpy().astype(np.float64))
        np.testing.assert_array_equal(mine, d[f"seed{seed}/{k}"], err_msg=k)  # bit-exact: same generator, same draws, same law


def test_conv_init_law():
    """std of every conv weight = sqrt(2/(kh*kw*Cout)), 1x1 convs included (reference mobilenet_v1.py:155-158)."""
    from trackertraincode.backbones.mobilenet_v1 import MobileNet

    torch.manual_seed(3)
    net = MobileNet(num_classes=None)
    for name, m in net.named_modules():
        if isinstance(m, torch.nn.Conv2d) and m.weight.numel() >= 4096:
            expect = (2.0 / (m.kernel_size[0] * m.kernel_size[1] * m.out_channels)) ** 0.5
            assert abs(float(m.weight.std()) / expect - 1.0) < 0.05, name
            assert abs(float(m.weight.mean())) < 0.1 * expect, name
