"""Data-parallel gradient exchange on CPU: 2 processes over gloo (the N>1 path of bench.py uses the same
GradAllReduce over RCCL)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, rdv_file, pkg, q):
    sys.path.insert(0, pkg)
    # file rendezvous: no TCP port to race for (the container hostname may not resolve either)
    dist.init_process_group("gloo", init_method=f"file://{rdv_file}", rank=rank, world_size=world)
    from trackertraincode.parallel import GradAllReduce, broadcast_module_state, shard_range

    torch.manual_seed(rank)
    model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    broadcast_module_state(model)
    params = list(model.parameters())
    g = torch.Generator().manual_seed(100 + rank)
    grads = [torch.randn(p.shape, generator=g) for p in params]
    red = GradAllReduce(bucket_bytes=64)  # tiny buckets: several collectives in flight
    # first two "become ready during backward" (handed over as returned tensors, then cloned into .grad)
    for p, gr in zip(params[2:], grads[2:]):
        red.on_ready([(p, gr)])
        p.grad = gr.clone()
    for p, gr in zip(params[:2], grads[:2]):
        p.grad = gr.clone()
    red.finish(params)
    # numpy, not tensors: torch shares tensors through /dev/shm handles that die with this process
    q.put((rank, [p.detach().numpy().copy() for p in params], [p.grad.numpy().copy() for p in params], shard_range(10, rank, world)))
    dist.destroy_process_group()


def test_two_rank_gradient_average_and_broadcast(tmp_path):
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neuralnet-tracker-traincode_amd")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rdv = str(tmp_path / "rendezvous")
    procs = [ctx.Process(target=_worker, args=(r, 2, rdv, pkg, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, w0, g0, s0), (_, w1, g1, s1) = res
    w0, w1, g0, g1 = ([torch.from_numpy(a) for a in x] for x in (w0, w1, g0, g1))
    for a, b in zip(w0, w1):
        assert torch.equal(a, b)  # broadcast from rank 0
    # expected average: regenerate both ranks' gradients in order
    gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]
    per_rank = [[torch.randn(p.shape, generator=g) for p in w0] for g in gens]
    for i in range(len(w0)):
        avg = (per_rank[0][i] + per_rank[1][i]) / 2
        assert torch.allclose(g0[i], avg, atol=1e-7) and torch.allclose(g1[i], avg, atol=1e-7)
    assert s0 == (0, 5) and s1 == (5, 10)
