"""`--precision`: fp32 (the reference's) and bf16-compute are the two modes of the MobileNet training kernels; the storage-only variants of
earlier rounds ("bf16", "bf16-all": bf16 tensors under the fp32 kernels, slower than fp32) are retired and must say so, naming the mode that
replaced them.  Precision is an attribute of the backbone instance and stays out of the checkpoint surface (reference io.py:19-27)."""
import pytest
import torch


def test_retired_modes_raise_and_name_the_replacement():
    import trackertraincode.backbones.mobilenet_v1 as MB

    for mode in ("bf16", "bf16-all", torch.bfloat16):
        with pytest.raises(ValueError, match="bf16-compute"):
            MB.set_activation_dtype(mode)
        with pytest.raises(ValueError, match="bf16-compute"):
            MB.MobileNet(num_classes=0).set_precision(mode)
    assert MB._DEFAULT_PRECISION == "fp32"
    with pytest.raises(ValueError):
        MB.set_activation_dtype("fp8")


def test_precision_is_per_instance_and_not_in_the_checkpoint():
    import trackertraincode.backbones.mobilenet_v1 as MB
    from trackertraincode.neuralnets.models import NetworkWithPointHead

    a, b = MB.MobileNet(num_classes=0), MB.MobileNet(num_classes=0).set_precision("bf16-compute")
    assert a.effective_precision() == "fp32" and b.effective_precision() == "bf16-compute"
    MB.set_activation_dtype("bf16-compute")
    try:
        assert a.effective_precision() == "bf16-compute" and a.set_precision("fp32").effective_precision() == "fp32"
    finally:
        MB.set_activation_dtype("fp32")
    assert b.effective_precision() == "bf16-compute" and b.set_precision(None).effective_precision() == "fp32"
    net = NetworkWithPointHead(enable_point_head=False)
    net.convnet.set_precision("bf16-compute")
    assert "precision" not in net.get_config() and not any("precision" in k for k in net.state_dict())
