"""The data-parallel step on real hardware with a world of ONE: RCCL process group (backend "nccl"), the backbone's
grad-ready hook, the second (weight-gradient) stream, bucketed asynchronous all-reduces on their own stream and
`finish()` - everything the N>1 path of bench.py runs, checked against the same step without any of it."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist

from util import build_net, load_golden, make_batches, script_args, train_script

pytestmark = pytest.mark.gpu


def test_single_rank_rccl_step_equals_plain_step(tmp_path):
    import trackertraincode.backbones.mobilenet_v1 as MB
    import trackertraincode.train as train
    from trackertraincode import parallel
    from trackertraincode.parallel import GradAllReduce, broadcast_module_state

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"file://{tmp_path / 'rdv'}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        d, meta = load_golden("model_default.npz")
        S = train_script()

        def run(with_reducer):
            net = build_net(meta, "cuda").train()
            crit, _ = S.setup_losses(script_args(meta["flags"]), net)
            opt, _ = S.create_optimizer(net, script_args(meta["flags"], epochs=20))
            broadcast_module_state(net)
            params = list(net.parameters())
            red = GradAllReduce(bucket_bytes=1 << 20, always_reduce=True) if with_reducer else None
            parallel.install(red)
            collectives = []
            try:
                losses, first_grads = [], None
                for it in range(3):
                    for p in params:
                        p.grad = None
                    if red:
                        red.begin_step()
                    out = train.training_step(net, make_batches(meta, "cuda"), 0, crit)
                    out["loss"].backward()
                    if red:
                        red.finish(params)
                        collectives.append(red.collectives)
                        # zero copy: the gradients autograd installed ARE the views of the arenas that travelled
                        assert red.copied == 0 and red.zero_copy >= 90, (red.copied, red.zero_copy)
                    if it == 0:  # later steps diverge chaotically at B=8 (atomics order -> Adam), see test_model_gpu
                        torch.cuda.synchronize()
                        first_grads = [p.grad.cpu().numpy().copy() for p in params]
                    opt.step()
                    losses.append(out["loss"].item())
                torch.cuda.synchronize()
            finally:
                parallel.install(None)
            if red:
                # 12.9 MB of backbone gradients in 1 MB buckets + the heads' arena + one packed bucket of stragglers:
                # O(buckets) collectives, not O(parameters) (the model has ~190 parameter tensors)
                assert all(5 <= c <= 20 for c in collectives), collectives
            return losses, [p.detach().cpu().numpy().copy() for p in params], first_grads

        l0, p0, g0 = run(False)
        l1, p1, g1 = run(True)
        np.testing.assert_allclose(l1[0], l0[0], rtol=1e-5)
        np.testing.assert_allclose(l1, l0, rtol=5e-3)
        for a, b in zip(g1, g0):  # first step: identical up to the order of the fp32 weight-gradient atomics
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-5 * max(1.0, float(np.abs(b).max())))
        lr = 1.0e-3
        for a, b in zip(p1, p0):
            np.testing.assert_allclose(a, b, rtol=1e-4, atol=8 * lr)  # atomics-order noise through Adam's normalised update
    finally:
        dist.destroy_process_group()
