"""Data-parallel gradient exchange on CPU: 2 processes over gloo (the N>1 path of bench.py uses the same GradAllReduce
over RCCL).  The arena protocol is driven with the REAL layout of the default backbone - 13 blocks of six tensors in
reverse layer order, the stem last, the heads' arena first - plus a few stragglers outside any arena."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

# (name, Cin, Cout): the 13 depthwise-separable blocks of reference backbones/mobilenet_v1.py:128-140
BLOCKS = [("dw2_1", 32, 64), ("dw2_2", 64, 128), ("dw3_1", 128, 128), ("dw3_2", 128, 256), ("dw4_1", 256, 256), ("dw4_2", 256, 512),
          ("dw5_1", 512, 512), ("dw5_2", 512, 512), ("dw5_3", 512, 512), ("dw5_4", 512, 512), ("dw5_5", 512, 512), ("dw5_6", 512, 1024),
          ("dw6", 1024, 1024)]


def backbone_shapes():
    shapes = [(32, 1, 5, 5), (32,), (32,)]
    for _, ci, co in BLOCKS:
        shapes += [(ci, 1, 3, 3), (ci,), (ci,), (co, ci, 1, 1), (co,), (co,)]
    return shapes


def _worker(rank, world, rdv_file, pkg, q):
    sys.path.insert(0, pkg)
    # file rendezvous: no TCP port to race for (the container hostname may not resolve either)
    dist.init_process_group("gloo", init_method=f"file://{rdv_file}", rank=rank, world_size=world)
    from trackertraincode.parallel import GradAllReduce, broadcast_module_state, shard_range

    torch.manual_seed(rank)
    model = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    broadcast_module_state(model)
    stragglers = list(model.parameters())  # gradients that come through no arena (packed by finish())

    # ---- the backbone's arena: 64-element aligned slots in parameter order, announced block by block from the back
    shapes = backbone_shapes()
    params = [torch.nn.Parameter(torch.zeros(s)) for s in shapes]
    offs, total = [], 0
    for prm in params:
        offs.append(total)
        total += (prm.numel() + 63) // 64 * 64
    offs.append(total)
    g = torch.Generator().manual_seed(100 + rank)
    arena = torch.zeros(total)
    for i, prm in enumerate(params):
        arena[offs[i]:offs[i] + prm.numel()] = torch.randn(prm.numel(), generator=g)
    local = arena.clone()
    # ---- the heads' arena (announced first: the heads' backward runs before the backbone's)
    NZ, F = 68, 1024
    hparams = [torch.nn.Parameter(torch.zeros(NZ, F)), torch.nn.Parameter(torch.zeros(NZ)), torch.nn.Parameter(torch.zeros(8, 4))]
    harena = torch.randn(NZ * F + NZ + 64, generator=g)
    hlocal = harena.clone()

    red = GradAllReduce(bucket_bytes=4 << 20)
    red.begin_step()
    red.on_ready(harena, [(hparams[0], 0, NZ * F), (hparams[1], NZ * F, NZ * F + NZ), (hparams[2], NZ * F + NZ, NZ * F + NZ + 32)])
    for k in range(len(BLOCKS) - 1, -1, -1):
        pi = 3 + 6 * k
        red.on_ready(arena, [(params[i], offs[i], offs[i + 1]) for i in range(pi, pi + 6)])
    red.on_ready(arena, [(params[i], offs[i], offs[i + 1]) for i in range(3)])
    # what autograd does after backward: installs the arena VIEWS as .grad
    for i, prm in enumerate(params):
        prm.grad = arena[offs[i]:offs[i] + prm.numel()].view(prm.shape)
    hparams[0].grad, hparams[1].grad = harena[:NZ * F].view(NZ, F), harena[NZ * F:NZ * F + NZ]
    hparams[2].grad = harena[NZ * F + NZ:NZ * F + NZ + 32].view(8, 4).clone()  # a CLONED gradient: finish() must repair it
    sg = [torch.randn(p.shape, generator=g) for p in stragglers]
    for p, gr in zip(stragglers, sg):
        p.grad = gr.clone()
    red.finish(params + hparams + stragglers)
    # numpy, not tensors: torch shares tensors through /dev/shm handles that die with this process
    q.put(dict(rank=rank, collectives=red.collectives, copied=red.copied, zero_copy=red.zero_copy, local=local.numpy().copy(), hlocal=hlocal.numpy().copy(),
               summed=arena.numpy().copy(), hsummed=harena.numpy().copy(), hp2=hparams[2].grad.numpy().copy(),
               w=[p.detach().numpy().copy() for p in stragglers], sg_local=[x.numpy().copy() for x in sg],
               sg=[p.grad.numpy().copy() for p in stragglers], shard=shard_range(10, rank, world), scale=red.grad_scale,
               same_storage=all(prm.grad.data_ptr() == arena.data_ptr() + 4 * offs[i] for i, prm in enumerate(params))))
    dist.destroy_process_group()


def test_two_rank_arena_exchange_in_place(tmp_path):
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "neuralnet-tracker-traincode_amd")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    rdv = str(tmp_path / "rendezvous")
    procs = [ctx.Process(target=_worker, args=(r, 2, rdv, pkg, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t["rank"])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0, r1 = res
    import numpy as np

    for a, b in zip(r0["w"], r1["w"]):
        assert np.array_equal(a, b)  # broadcast from rank 0
    # sums over the two replicas, in place, in both arenas (the division by the world size is the optimiser's grad_scale)
    for r in res:
        np.testing.assert_allclose(r["summed"], r0["local"] + r1["local"], rtol=0, atol=1e-6)
        n = 68 * 1024 + 68 + 32
        np.testing.assert_allclose(r["hsummed"][:n], (r0["hlocal"] + r1["hlocal"])[:n], rtol=0, atol=1e-6)
        np.testing.assert_allclose(r["hp2"].reshape(-1), (r0["hlocal"] + r1["hlocal"])[68 * 1024 + 68:n], rtol=0, atol=1e-6)  # repaired clone
        for i in range(len(r["sg"])):
            np.testing.assert_allclose(r["sg"][i], r0["sg_local"][i] + r1["sg_local"][i], rtol=0, atol=1e-6)
        assert r["same_storage"] and r["copied"] == 1 and r["zero_copy"] == 81 + 2
        assert r["scale"] == 0.5
        # 12.8 MB of backbone gradients in >= 4 MB buckets (reverse layer order: dw6 alone is 4.2 MB) + the heads' arena
        # + one packed bucket of stragglers: a handful of collectives for 88 tensors
        assert 4 <= r["collectives"] <= 7, r["collectives"]
    assert r0["shard"] == (0, 5) and r1["shard"] == (5, 10)
