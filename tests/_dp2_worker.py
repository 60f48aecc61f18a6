"""Worker of tests/test_dp2_gpu.py: one of TWO data-parallel replicas of the real training step, both on the one GPU of the box,
exchanging gradients through torch.distributed's `gloo` backend (BASELINE config 4's semantics - two replicas, per-replica
BatchNorm, in-place arena all-reduce during backward, 1/world folded into clip+Adam - minus RCCL/xGMI, which need two GPUs).

Each rank: real backward hooks (mobilenet_v1.grad_ready_hook, _hipops.grad_ready_hook), real arenas, GradAllReduce buckets,
`finish()`, `ClipAdam.grad_scale = 1/2`, one optimiser step.  Rank 0 then computes the expectation in the same process
without any of it: two single-replica backward passes (its own crops, the other rank's crops), gradients averaged by hand,
the same optimiser step.  Prints "RESULT <json>".
usage: _dp2_worker.py <repo> <rank> <world> <port> <B per rank>
"""
import hashlib
import json
import os
import sys

repo, rank, world, port, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
for p_ in (repo, repo + "/neuralnet-tracker-traincode_amd", repo + "/tests"):
    sys.path.insert(0, p_)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from util import build_net, load_golden, make_batches, script_args, train_script  # noqa: E402
import trackertraincode.train as train  # noqa: E402
from trackertraincode import parallel  # noqa: E402
from trackertraincode.parallel import GradAllReduce, broadcast_module_state  # noqa: E402

dist.init_process_group("gloo", rank=rank, world_size=world)
_, meta = load_golden("model_full.npz")
S = train_script()
EPOCH = 150


def meta_of(r):  # rank r's crops: its own seed
    return dict(meta, B=B, split=(B * 5) // 8, input_seed=meta["input_seed"] + 17 * r)


def fresh():
    net = build_net(meta, "cuda").train()
    crit, _ = S.setup_losses(script_args(meta["flags"]), net)
    opt, _ = S.create_optimizer(net, script_args(meta["flags"], epochs=20))
    for g in opt.param_groups:  # the full learning rates (the schedule's warm-up would start at a tenth)
        g["lr"] = g["lr"] * 10.0
    return net, crit, opt


# does this torch's gloo take CUDA tensors?  Otherwise the exchange is staged through the host (test-only).
probe = torch.ones(4, device="cuda")
try:
    dist.all_reduce(probe)
    torch.cuda.synchronize()
    mode = "gloo on CUDA tensors" if float(probe[0]) == world else "host-staged"
except Exception:  # noqa: BLE001
    mode = "host-staged"
if mode == "host-staged":
    def _host_all_reduce(self, flat):
        torch.cuda.current_stream().synchronize()
        h = flat.detach().cpu()
        dist.all_reduce(h, group=self.pg)
        flat.copy_(h)
        self.collectives += 1
    GradAllReduce._all_reduce = _host_all_reduce

# ---- the data-parallel step
net, crit, opt = fresh()
broadcast_module_state(net)
params = list(net.parameters())
red = GradAllReduce(bucket_bytes=1 << 20)
assert red.world == world and red.active
parallel.install(red)
try:
    opt.zero_grad(set_to_none=True)
    red.begin_step()
    out = train.training_step(net, make_batches(meta_of(rank), "cuda"), EPOCH, crit)
    out["loss"].backward()
    red.finish(params)
    opt.grad_scale = red.grad_scale
    summed = [None if p.grad is None else p.grad.detach().clone() for p in params]
    opt.step()
    torch.cuda.synchronize()
finally:
    parallel.install(None)
state = {k: v.detach().cpu() for k, v in net.state_dict().items()}
h = hashlib.sha256()
for p in params:
    h.update(p.detach().cpu().numpy().tobytes())
hashes = [None] * world
dist.all_gather_object(hashes, h.hexdigest())
res = dict(rank=rank, mode=mode, collectives=red.collectives, zero_copy=red.zero_copy, copied=red.copied, loss=out["loss"].item(),
           replicas_in_sync=len(set(hashes)) == 1)

if rank == 0:
    # ---- expectation: single-replica passes, averaged by hand
    gs, ref_net, ref_opt = [], None, None
    for r in range(world):
        n2, c2, o2 = fresh()
        o2.zero_grad(set_to_none=True)
        train.training_step(n2, make_batches(meta_of(r), "cuda"), EPOCH, c2)["loss"].backward()
        torch.cuda.synchronize()
        gs.append([None if p.grad is None else p.grad.detach().clone() for p in n2.parameters()])
        if r == 0:
            ref_net, ref_opt = n2, o2
    worst_sum = 0.0
    for i, p in enumerate(ref_net.parameters()):
        if gs[0][i] is None:
            assert summed[i] is None
            continue
        tot = sum(g[i] for g in gs)
        d = float((summed[i] - tot).abs().max()) / max(float(tot.abs().max()), 1e-30)
        worst_sum = max(worst_sum, d)
        p.grad = tot / world
    ref_opt.step()
    torch.cuda.synchronize()
    par_abs, run_rel, worst_key, bitwise = 0.0, 0.0, "", True
    pnames = {k for k, _ in ref_net.named_parameters()}
    for k, v in ref_net.state_dict().items():
        a, b = state[k].double(), v.detach().cpu().double()
        if not torch.equal(a, b):
            bitwise = False
        if k in pnames:
            d = float((a - b).abs().max())
            if d > par_abs:
                par_abs, worst_key = d, k
        else:
            run_rel = max(run_rel, float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30))
    res.update(grad_sum_rel=worst_sum, param_abs=par_abs, worst_key=worst_key, buffers_rel=run_rel, bitwise=bitwise, n_params=len(params),
               lr=ref_opt.param_groups[0]["lr"])
print("RESULT " + json.dumps(res))
dist.barrier()
dist.destroy_process_group()
