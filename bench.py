#!/usr/bin/env python3
"""Benchmark of the pose-estimator training step on MI355X (BASELINE.json metric:
face-crops/sec forward+backward at per-GPU batch 512).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = zero_grad + forward + multi-task loss + backward + fused clip/Adam step of
NetworkWithPointHead("mobilenetv1") with the training script's default flags (landmark head on) on one
synthetic batch that is already resident in HBM (+ the RCCL gradient all-reduce, overlapped with
backward, when N > 1; weak scaling: 512 crops per GPU).  The optimiser step is INSIDE `value`;
`optimizer_ms` reports it alone for reference.  `--backbone resnet18` / `--batch 256` time BASELINE
configs 3 and 2 the same way.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
for p in (REPO, os.path.join(REPO, "neuralnet-tracker-traincode_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: dense bf16 / fp16 MFMA peak (v_mfma_f32_32x32x16_{bf16,f16}, 32 cycles per issue)
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0          # HBM3E spec (6.29 TB/s measured copy)
ALGO_BYTES_PER_CROP = 55.55e6  # 13 888 321 fp32 elements (bf16 activation storage: 27.78 MB)
ALGO_FLOP_PER_CROP = 1.372e9   # 686 045 536 MAC fwd+bwd


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="crops per GPU")
    ap.add_argument("--backbone", default="mobilenetv1", choices=["mobilenetv1", "resnet18"])
    ap.add_argument("--blurpool", action="store_true", help="the training script's --blurpool: BlurPool2D + stride-1 depthwise conv in the strided MobileNet blocks (ResNet18: in front of every "
                    "block's first convolution and in the max-pool's place)")
    ap.add_argument("--precision", default="fp32",
                    help="fp32 (default, the headline) | bf16-compute = BASELINE config 5's bf16 leg (separate line, dtype bf16; mobilenetv1 only): activations "
                    "and gradients bf16 in 64-channel blocks, pointwise convolutions as ONE bf16 MFMA product with fp32 accumulation (csrc/bc_*.hip).  The "
                    "storage-only variants bf16 / bf16-all of earlier rounds are retired and raise")
    ap.add_argument("--no-legs", action="store_true", help="skip the two short extra legs the default N = 1 run appends to its line (`legs`: batch 256 fp32 = "
                    "BASELINE config 2, batch 512 bf16-compute = config 5's bf16 leg; 10 steps each after the headline is measured)")
    ap.add_argument("--traffic-json", default=None, help="rocprofv3 PMC summary (tools/pmc_summary.py) taken with THIS build; "
                    "fills roofline.traffic (null without it)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL over xGMI (the measurement); gloo with --share-gpu: "
                    "a dry run of the multi-rank code path on a one-GPU box (tests/test_bench_dp_dryrun_gpu.py) - its rate is not a result")
    ap.add_argument("--share-gpu", action="store_true", help="dry run: every rank uses device 0")
    ap.add_argument("--comm-only", action="store_true", help="N > 1: time ONLY the bucketed in-place gradient all-reduce of one step (the buckets a real "
                    "backward announces, replayed on buffers of the same sizes): ms per step and bus bandwidth, no compute beside it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-copy-probe", action="store_true", help="skip the same-process streaming probe (copy_probe object)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-call HIP-event instrumentation (roofline = null)")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every kernel from Python instead of replaying one captured hipGraph")
    ap.add_argument("--force-graph", action="store_true", help="always replay the captured hipGraph (default at N=1: whichever of graph replay "
                    "and eager enqueue the warm-up measures faster)")
    ap.add_argument("--serial-streams", action="store_true",
                    help="(kept for old command lines; every kernel of a step already runs on one stream - the weight-gradient side stream of earlier "
                    "rounds is an experiment hook that is off)")
    ap.add_argument("--per-call", action="store_true", help="print the roofline pass call by call (kernel, shape, us, GB/s, TFLOP/s) on stderr")
    ap.add_argument("--cpu-batch", type=int, default=64)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="time budget of the CPU baseline sample")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = min(host cores, 32)")
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of the C-ABI calls on the stream they are launched on (torch's current stream),
    with the algorithmic flops/bytes of each call."""

    act_bytes = 4  # bytes per activation element in HBM (2 on the bf16-compute path)

    def __init__(self):
        self.records = []  # (name, start, end, flops, bytes)
        self.enabled = False

    @staticmethod
    def work(name, a):
        """(kernel name as rocprofv3 prints it, flops, algorithmic bytes) of one call; fp32 = 4 B/element.
        Mirrors the dispatch rules of csrc/pwconv.hip + pwconv_f16.hip + pwconv_r.hip + dwconv_tiled.hip (the A/B environment switches
        mentioned below exist in experiment builds of the library only)."""
        ints = [x for x in a if isinstance(x, int) and not isinstance(x, bool)]
        eb = KernelTimer.act_bytes
        if name == "ttk_pwconv1x1_bwd_fused":  # first two pointwise layers: weight + data gradient, operands read once (HBM-bound)
            M, ci, co = ints[-3:]
            f16 = ci == 128 or (ci == 64 and os.environ.get("TTK_GEMM") not in ("f32mfma", "bf16x3") and not os.environ.get("TTK_FUSED_FP32"))  # csrc/pw_bwd_fused.hip: fused_f16
            return f"pw_bwd_fused{'16' if f16 else ''}_k<{ci}, {co}>", 4 * M * ci * co, 4 * (2 * M * co + 2 * M * ci) + 4 * ci * co
        if name.startswith("ttk_bc_pw"):  # bf16-compute path: one bf16 product, every activation-sized tensor 2 bytes per element
            M, ci, co = ints[-3:]
            fl = 2 * M * ci * co
            if name == "ttk_bc_pw_fwd":
                l128 = co == 128 and ci == 256  # csrc/bc_gemm.hip: is_l128 (N = 128, K = 256 runs the streamed kernel on 128-channel tiles)
                return ("bc_gemm_l_k<0, 128>" if l128 else f"bc_gemm_e_k<0, {co // 32}, {ci // 32}>" if (co <= 128 and ci <= 256) else "bc_gemm_l_k<0, 256>"), fl, 2 * (M * ci + M * co) + 2 * ci * co
            if name == "ttk_bc_pw_bwd_data":
                l128 = ci == 128 and co == 256
                return ("bc_gemm_l_k<1, 128>" if l128 else f"bc_gemm_e_k<1, {ci // 32}, {co // 32}>" if (ci <= 128 and co <= 256) else "bc_gemm_l_k<1, 256>"), fl, 2 * (2 * M * co + 2 * M * ci) + 2 * ci * co
            if name == "ttk_bc_pw_bwd_fused":  # weight AND data gradient of the early layers, operands read once
                return f"bc_bwd_fused_k<{co // 32}, {ci // 32}", 2 * fl, 2 * (2 * M * co + 2 * M * ci) + 6 * ci * co
            tn, tk = min(co, 256) // 32, min(ci, 256) // 32
            return f"bc_wgrad_k<{tn}, {tk}", fl, 2 * (2 * M * co + M * ci) + 4 * ci * co
        if name in ("ttk_bc_dw_fwd", "ttk_bc_dw_bwd_data"):
            B, H, W, C, s_ = a[-5:]
            n_in, n_out = B * H * W * C, B * ((H - 1) // s_ + 1) * ((W - 1) // s_ + 1) * C
            tf = lambda v: "true" if v else "false"
            if name == "ttk_bc_dw_fwd":  # bc_dw_fwd_k<stride, residual producer, channels per block, band carry> (the last from the tiling: not named)
                return f"bc_dw_fwd_k<{s_}, {tf(a[2])}, {min(C, 64)}", 2 * 9 * n_out, 2 * (n_in * (1 + bool(a[2]) + bool(a[3])) + n_out)
            lean = not (a[4] or a[7] or a[8])  # csrc/bc_dw.hip: no residual gradient, no residual producer, no materialised block input
            return f"bc_dw_bwd_k<{s_}, {min(C, 64)}, {tf(lean)}>", 2 * 2 * 9 * n_out, 2 * (2 * n_out + n_in * (2 + bool(a[4]) + bool(a[7] or a[8])))
        if name.startswith("ttk_pwconv1x1"):
            # trailing arguments: ..., M, Cin, Cout, [scratch pointer of the split weights,] act_bf16
            M, ci, co = ints[-4:-1] if name == "ttk_pwconv1x1_bwd_weight" else ints[-5:-2]
            fl = 2 * M * ci * co
            if name == "ttk_pwconv1x1_fwd":
                K, N, mode, by = ci, co, 0, eb * (M * ci + M * co) + 4 * ci * co
            elif name == "ttk_pwconv1x1_bwd_data":
                K, N, mode, by = co, ci, 1, eb * (2 * M * co + 2 * M * ci) + 4 * ci * co
            else:
                by = eb * (2 * M * co + M * ci) + 4 * ci * co
                t_mode = os.environ.get("TTK_WGRAD_T", "u")[:1]  # csrc/pwconv_r.hip f16t_wgrad_shape: 0 | t (256 x 256 tiles) | u (128 x 256, default)
                if t_mode != "0" and ci % 256 == 0 and ci >= 256 and os.environ.get("TTK_GEMM") in (None, "", "f16x2"):
                    ty = "float" if eb == 4 else "unsigned short"
                    if t_mode == "t" and co % 256 == 0:
                        return f"pw16t_wgrad_k<{ty}>", fl, by  # 256 x 256 tiles, transposed LDS reads
                    if t_mode != "t" and co % 128 == 0 and not (ci == 256 and co == 256):
                        return f"pw16u_wgrad_k<{ty}>", fl, by  # 128 x 256 tiles, eight producer waves with two register sets
                if ci >= 128 and co >= 128 and (ci % 256 == 0 or co % 256 == 0 or (ci == 128 and co == 128)):
                    tn = "float" if eb == 4 else "unsigned short"
                    return (f"pw16_wgrad_k<128, 256, 1, {tn}>" if ci % 256 == 0 else f"pw16_wgrad_k<256, 128, 1, {tn}>" if co % 256 == 0 else f"pw16_wgrad_k<128, 128, 1, {tn}>"), fl, by
                return "pw_wgrad_k", fl, by
            if K >= 128 and N % 256 == 0:
                if os.environ.get("TTK_GEMM_R", "1") != "0" and os.environ.get("TTK_GEMM") in (None, "", "f16x2") and not (mode == 1 and K == 256 and N == 256):
                    # row-block kernel (csrc/pwconv_r.hip: f16r_gemm_shape, r_plan): one partial-sum row per row block of rt rows, tile = 32 rblk rows
                    # (round 4: the data gradient's 128- and 192-row tiles run the eight-wave form pw16m_k)
                    import trackertraincode._hip as H
                    rblk = H.lib().cdll.ttk_pwconv_tile_rows(M, K, N, mode) // 32
                    kern = "pw16m_k" if (mode == 1 and rblk in (4, 6)) else "pw16r_k"
                    return f"{kern}<{rblk}, {mode}, {'float' if eb == 4 else 'unsigned short'}>", fl, by
                return f"pw16_k<128, 256, {mode}, {mode}, 1, {'float' if eb == 4 else 'unsigned short'}>", fl, by  # <BM, BN, A form, epilogue form, register sets, storage>
            if K >= 64 and N == 128:
                return f"pw16_k<256, 128, {mode}, {mode}, 1, {'float' if eb == 4 else 'unsigned short'}>", fl, by
            if N == 64 and K == 32:
                return f"pw_gemm_k<64, 2, 2, {mode}, 1>", fl, by  # single LDS stage
            return (f"pw_gemm_k<{min(N, 128)}, 2, 2, {mode}, 2>" if N >= 64 else f"pw_gemm_k<32, 4, 1, {mode}, 2>"), fl, by
        if name == "ttk_dwconv3x3_fwd":
            B, H, W, C, s_ = a[-6:-1]
            n_in, n_out = B * H * W * C, B * ((H - 1) // s_ + 1) * ((W - 1) // s_ + 1) * C
            by = eb * (n_in * (1 + bool(a[2]) + bool(a[3])) + n_out)
            return f"dw_fwd_tiled_k<{s_}, {'float' if eb == 4 else 'unsigned short'}>", 2 * 9 * n_out, by
        if name == "ttk_dwconv3x3_bwd_data":
            B, H, W, C, s_ = a[-6:-1]
            n_in, n_out = B * H * W * C, B * ((H - 1) // s_ + 1) * ((W - 1) // s_ + 1) * C
            by = eb * (2 * n_out + n_in * (2 + bool(a[4]) + bool(a[7] or a[8])))
            return f"dw_bwd_tiled_k<{s_}, {'float' if eb == 4 else 'unsigned short'}>", 2 * 2 * 9 * n_out, by
        if name in ("ttk_conv_fwd", "ttk_conv_bwd_data", "ttk_conv_bwd_weight"):  # ResNet18 implicit GEMMs
            B, H, W, ci, co, kh, kw, s_, _pad = ints[-9:]
            Ho, Wo = (H - 1) // s_ + 1, (W - 1) // s_ + 1
            fl = 2 * B * Ho * Wo * ci * co * kh * kw
            n_in, n_out = B * H * W * ci, B * Ho * Wo * co
            by = 4 * {"ttk_conv_fwd": n_in + n_out, "ttk_conv_bwd_data": 2 * n_out + 2 * n_in, "ttk_conv_bwd_weight": 2 * n_out + n_in}[name]
            kern = "pw_split_k" if os.environ.get("TTK_GEMM") == "bf16x3" else "pw16_k"  # 6 | 3 sixteen-bit MFMA products per fp32 product
            return name.replace("ttk_", "") + f" ({kern} implicit GEMM)", fl, by
        return name, 0, 0

    def wrap(self, lib):
        orig = lib.call
        timer = self
        timer.shapes = []

        def call(name, *args):
            # only the conv kernels (94 % of the GPU time) are bracketed, and only in the separate roofline pass
            if not timer.enabled or not (name.startswith("ttk_pwconv1x1") or name.startswith("ttk_dwconv3x3") or name.startswith("ttk_conv_") or name.startswith("ttk_bc_pw") or name.startswith("ttk_bc_dw")):
                return orig(name, *args)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            orig(name, *args)
            e.record()
            kern, fl, by = timer.work(name, args)
            ints = [x for x in args if isinstance(x, int) and not isinstance(x, bool)]
            timer.records.append((kern, s, e, fl, by))
            timer.shapes.append(tuple(ints[-3:]) if (name == "ttk_pwconv1x1_bwd_fused" or name.startswith("ttk_bc_pw")) else tuple(args[-5:]) if name.startswith("ttk_bc_dw") else tuple(ints[-5:-2]) if name in ("ttk_pwconv1x1_fwd", "ttk_pwconv1x1_bwd_data")
                                else tuple(ints[-4:-1]) if name == "ttk_pwconv1x1_bwd_weight"
                                else tuple(ints[-9:]) if name.startswith("ttk_conv_") else tuple(args[-6:-1]))

        lib.call = call

    def per_call(self, steps):
        """Average over the steps of the pass, one line per call position inside a step."""
        n = len(self.records) // steps
        lines = []
        for i in range(n):
            recs = [self.records[i + k * n] for k in range(steps)]
            us = sum(s.elapsed_time(e) for _, s, e, _, _ in recs) / steps * 1e3
            kern, _, _, fl, by = recs[0]
            lines.append(f"{kern:38s} {str(self.shapes[i]):28s} {us:8.1f} us {by / us / 1e3:7.0f} GB/s {fl / us / 1e6:7.1f} TFLOP/s")
        return lines

    def summary(self, steps):
        agg = {}
        for name, s, e, fl, by in self.records:
            a = agg.setdefault(name, [0.0, 0, 0, 0])
            a[0] += s.elapsed_time(e)
            a[1] += 1
            a[2] += fl
            a[3] += by
        return {k: {"ms_per_step": v[0] / steps, "calls_per_step": v[1] / steps, "flops": v[2], "bytes": v[3], "ms": v[0]} for k, v in agg.items()}


def build_step(args, device):
    import numpy as np

    import trackertraincode.train as train
    from trackertraincode.neuralnets.models import NetworkWithPointHead
    from trackertraincode.pipelines import SyntheticPoseLoader, Tag

    import importlib.util

    spec = importlib.util.spec_from_file_location("amd_train_script", os.path.join(REPO, "neuralnet-tracker-traincode_amd", "scripts", "train_poseestimator.py"))
    S = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(S)

    def script_args(flags):
        ns = S.make_parser().parse_args([])
        for k, v in flags.items():
            setattr(ns, k, v)
        return ns

    torch.manual_seed(getattr(args, "seed", 0))  # (tools/soak.py --seed: the weight initialisation of its legs)
    net = NetworkWithPointHead(enable_point_head=True, enable_uncertainty=False, config=args.backbone,
                               backbone_args={"use_blurpool": args.blurpool})
    g = torch.Generator().manual_seed(7)  # synthetic 3DMM keypoint basis (the real blob is not in the reference)
    net.landmarks.deformablekeypoints.set_basis(torch.randn(68, 3, generator=g) * 0.5, torch.randn(50, 68, 3, generator=g) * 0.05)
    net = net.to(device).train()
    if args.backbone == "mobilenetv1":
        net.convnet.set_precision(args.precision)  # an attribute of THIS backbone: the extra legs build their own networks beside it
    flags = dict(with_pointhead=True, with_nll_loss=False, rampup_nll_losses=False)
    crit, _ = S.setup_losses(script_args(flags), net)
    opt, _ = S.create_optimizer(net, script_args(flags))
    rank = int(os.environ.get("RANK", 0))
    # mix of the reference's default training set (pipelines.py:399-453): landmark-labelled crops with and
    # without shape parameters
    loader = SyntheticPoseLoader(args.batch, [(Tag.POSE_WITH_LANDMARKS, 110.0), (Tag.POSE_WITH_LMKS_NO_SHAPE_PARAMS, 10.0)],
                                 device=device, seed=1234 + rank)
    batches = next(iter(loader))
    return net, crit, opt, batches, train


def copy_probe(device):
    """What THIS box's memory system streams, measured in this process right after the timed region (ttk_stream_probe,
    csrc/probe.hip): a 1 GB read-only sweep, a 1:1 copy and the depthwise backward's 5:1 read:write mix, each as a linear
    sweep and in the 128-byte-slab shape of the depthwise kernels (a 32-channel slab of a 512-channel channels-last tensor),
    16 B per lane, >= 72 KiB in flight per CU.  Best of a small sweep (loads in flight, workgroups per CU, cache policy),
    each timed with HIP events over 3 launches after one warm-up.  GB/s of bytes read + written."""
    import trackertraincode._hip as H
    L, p = H.lib(), H.ptr
    n = 1 << 28  # floats: 1 GiB
    src = torch.empty(n, dtype=torch.float32, device=device).normal_()
    dst = torch.empty(n, dtype=torch.float32, device=device)
    sink = torch.zeros(4, dtype=torch.float32, device=device)
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    out = {}
    for label, nread, nwrite in (("read", 1, 0), ("copy", 1, 1), ("mix5to1", 5, 1)):
        per = (n // nread) // 1024 * 1024          # floats per stream
        for shape, row_bytes, seg in (("linear", 4096, 4096), ("slab128", 2048, 128)):
            rows = per * 4 // row_bytes
            best = (0.0, None)
            for unroll in (4, 8):
                for bpc in (4, 8):
                    for nt in (1, 0):
                        if unroll * nread > 40:
                            continue
                        def launch():
                            L.call("ttk_stream_probe", p(src), p(dst) if nwrite else None, p(sink), rows, row_bytes, seg, nread, nwrite, per * 4, unroll, nt,
                                   cus * bpc)
                        launch()
                        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        s0.record()
                        for _ in range(3):
                            launch()
                        s1.record()
                        s1.synchronize()
                        gbs = 3 * (nread + nwrite) * per * 4 / (s0.elapsed_time(s1) * 1e-3) / 1e9
                        if gbs > best[0]:
                            best = (gbs, f"{unroll * nread} x 16 B per lane in flight, {bpc} workgroups per CU, {'non-temporal' if nt else 'plain'} loads")
            out[f"{label}_{shape}_GBs"] = round(best[0], 1)
            out[f"{label}_{shape}_config"] = best[1]
    out["note"] = ("ttk_stream_probe on this device: bytes read + written per second; linear = contiguous sweep, slab128 = 128-byte pieces 2048 B apart "
                   "(the depthwise kernels' 32-channel slab at C = 512); 1 GiB per measurement")
    return out


def run_leg(args, device, batch, precision, steps=10, warmup=3):
    """One short extra leg in the same process, after the headline is measured: its own network, optimiser and batch (precision is an
    attribute of the backbone instance), the whole step replayed as one hipGraph (eager enqueue in-process if the capture fails), `steps`
    timed steps between synchronisations.  Returns {"value", "ms_per_step", ...}."""
    import argparse as _ap
    import gc

    a = _ap.Namespace(**vars(args))
    a.batch, a.precision = batch, precision
    net, crit, opt, batches, train = build_step(a, device)
    params = list(net.parameters())

    def eager():
        for p in params:
            p.grad = None
        out = train.training_step(net, batches, 0, crit)
        out["loss"].backward()
        opt.step()
        return out["loss"]

    mode = "hipGraph replay"
    try:
        graphed = train.GraphedTrainStep(net, crit, opt)
        step = lambda: graphed.run(batches, 0)["loss"]
        step()
    except Exception as exc:  # noqa: BLE001  (a failed capture must not cost the leg: eager enqueue always works, in this process)
        print(f"bench.py: leg B={batch} {precision}: hipGraph capture failed ({type(exc).__name__}: {exc}); eager enqueue", file=sys.stderr)
        graphed, step, mode = None, eager, "eager Python launches"
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"value": batch * steps / dt, "unit": "crops/s", "ms_per_step": dt / steps * 1e3, "steps": steps, "warmup": warmup, "per_gpu_batch": batch,
           "dtype": "f32" if precision == "fp32" else "bf16", "precision": precision, "enqueue": mode, "loss": float(loss.item())}
    del graphed, net, opt, batches, params
    gc.collect()
    torch.cuda.empty_cache()
    return res


def cpu_baseline(args):
    """The CPU oracle (port of the reference's algorithm, pinned to reference goldens) timed on this
    box's host cores on a bounded sample of the same workload."""
    from oracle import refmodel as R
    from oracle.synth import make_inputs, make_labels, make_state

    # torch's CPU kernels stop scaling (and thrash) far below the 256 hardware threads of the GPU box's
    # host: use at most 32 threads and say so in `cores`.
    n = args.cpu_threads or min(os.cpu_count() or 1, 32)
    torch.set_num_threads(n)
    if args.backbone == "resnet18":  # the heads on 512 features + the ResNet-18 backbone under "convnet."
        heads = {k: v for k, v in R.state_shapes(True, False, num_features=512).items() if not k.startswith("convnet.")}
        shapes = {**R.resnet18_state_shapes(prefix="convnet.", use_blurpool=args.blurpool), **heads}
    else:
        shapes = R.state_shapes(True, False, use_blurpool=args.blurpool)
    st = R.state_from_numpy(make_state(shapes, 0))
    gmm = R.ShapeGmm(os.path.join(REPO, "tests", "golden", "shapeparams_gmm.npz"))
    crit, _ = R.setup_losses(with_pointhead=True, with_nll_loss=False, gmm=gmm)
    cfg = dict(enable_point_head=True, enable_uncertainty=False, config=args.backbone)

    def sample(B, seconds, max_steps):
        image, ids = make_inputs(B, seed=1, structured=False)
        lab = make_labels(B, seed=1)
        batch = [dict(tag="POSE_WITH_LANDMARKS", n=B, **{k: torch.from_numpy(v) for k, v in lab.items() if k != "dataset_weight"})]
        x, idt = torch.from_numpy(image), torch.from_numpy(ids)

        def step():
            for v in st.values():
                v.grad = None
            out, _ = R.network_forward(st, x, idt, cfg, True)
            loss, _ = R.compute_loss(out, batch, 0, crit)
            loss.backward()

        tw = time.perf_counter()
        step()  # warm-up (also tells how long one step takes)
        tw = time.perf_counter() - tw
        steps = 0
        t0 = time.perf_counter()
        while True:
            step()
            steps += 1
            dt = time.perf_counter() - t0
            if dt + tw > seconds or steps >= max_steps:
                break
        return {"batch": B, "value": B * steps / dt, "steps": steps, "seconds": round(dt, 2)}

    # the CPU path at B = 256 (BASELINE config 2's size) and at B = 64, inside one ~20 s budget
    small = sample(args.cpu_batch, 0.4 * args.cpu_seconds, 50)
    big = sample(256, 0.6 * args.cpu_seconds, 5) if args.cpu_batch != 256 else small
    return {"value": big["value"], "unit": "crops/s", "cores": n, "host_cores": os.cpu_count(), "kind": "port", "samples": [small, big],
            "sample": f"{big['steps']} fwd+bwd steps of the same network at batch 256 in {big['seconds']} s (value), and {small['steps']} at batch "
                      f"{small['batch']} ({small['value']:.1f} crops/s); fp32, torch CPU kernels, {n} of the host's {os.cpu_count()} hardware threads"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if (args.dist_backend != "nccl") != args.share_gpu:
        raise SystemExit("--dist-backend gloo and --share-gpu belong together (dry run of the multi-rank path on one GPU)")
    local = 0 if args.share_gpu else local
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"

    import trackertraincode._hip as H
    import trackertraincode.backbones.mobilenet_v1 as MB
    from trackertraincode import parallel
    from trackertraincode.parallel import GradAllReduce, broadcast_module_state

    MB._check_precision(args.precision)  # (raises for the retired storage-only modes, naming bf16-compute)
    KernelTimer.act_bytes = 4 if args.precision == "fp32" else 2
    net, crit, opt, batches, train = build_step(args, device)
    broadcast_module_state(net)
    reducer = GradAllReduce() if world > 1 else None
    if reducer is not None:
        parallel.install(reducer)            # gradient arenas are all-reduced in place while backward continues
        opt.grad_scale = reducer.grad_scale  # 1/world, applied inside the fused clip+Adam kernel
    params = list(net.parameters())

    graphed = train.GraphedTrainStep(net, crit, opt) if (world == 1 and not args.no_graph) else None

    def eager_step():
        for p in params:
            p.grad = None
        out = train.training_step(net, batches, 0, crit)
        out["loss"].backward()
        if reducer is not None:
            reducer.finish(params)
        opt.step()  # fused global-norm clip + Adam (2 launches, no host sync)
        return out["loss"]

    use_graph = [graphed is not None]

    def step():
        # single GPU: the whole step (zero_grad, forward, losses, backward, clip+Adam) can be one captured hipGraph that
        # is replayed - same kernels, one enqueue call instead of ~200; data-parallel runs stay eager (RCCL calls
        # are issued from Python between backward and the optimiser)
        if use_graph[0] and not timer.enabled:
            return graphed.run(batches, 0)["loss"]
        return eager_step()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.serial_streams:
        MB._USE_WGRAD_STREAM = False
    timer = KernelTimer()
    timer.wrap(H.lib())
    enqueue_note = ""
    if graphed is not None and not args.force_graph:
        # Both enqueue modes run the same kernels.  Replay removes the host from the loop; eager enqueue lets the
        # launch-ahead queue hide the gaps between the ~200 small dependent kernels.  Which one wins depends on the host:
        # measure both during warm-up (untimed, runs as long as the timed region) and use the faster.
        probe = {}
        for mode in (False, True):  # eager first: it then runs exactly as it would without any capture in the process
            use_graph[0] = mode
            try:
                for _ in range(max(args.warmup, 2)):
                    step()
                sync()
                tp = time.perf_counter()
                for _ in range(max(args.steps, 4)):
                    step()
                sync()
                probe[mode] = (time.perf_counter() - tp) / max(args.steps, 4) * 1e3
            except Exception as exc:  # a failed capture must not cost the measurement: eager enqueue always works
                if not mode:
                    raise
                probe[mode] = float("inf")
                print(f"bench.py: hipGraph capture failed ({type(exc).__name__}: {exc}); using eager enqueue", file=sys.stderr)
        use_graph[0] = probe[True] <= probe[False]
        enqueue_note = f"; warm-up probe: graph replay {probe[True]:.2f} ms/step, eager {probe[False]:.2f} ms/step"
        if not use_graph[0]:
            # a live capture slows later eager steps by ~1 ms (its private memory pool): drop it
            import gc
            graphed.graph = graphed._out = None
            graphed._static = []
            for p in params:
                p.grad = None
            gc.collect()
            torch.cuda.empty_cache()
        for _ in range(2):
            step()
    else:
        for _ in range(args.warmup):
            step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    host_enqueue = time.perf_counter() - t0  # host time to ENQUEUE the steps (no sync yet)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # Per-kernel pass for the roofline object: the same step again with HIP events around every conv C-ABI call (one stream, as in the timed
    # region).  It is a separate pass because the events cost ~3 ms/step of host time, which would distort `value`.
    roof_steps = 0 if args.no_kernel_timing else min(args.steps, 10)
    if roof_steps:
        use_side = MB._USE_WGRAD_STREAM
        MB._USE_WGRAD_STREAM = False
        step()
        sync()
        timer.enabled = True
        for _ in range(roof_steps):
            step()
        sync()
        timer.enabled = False
        MB._USE_WGRAD_STREAM = use_side
    # Communication diagnostics (N > 1; a separate pass: the extra events and stream waits are not free): per step the buckets and bytes the
    # reducer exchanged, the device time of the collectives (they overlap backward), and the EXPOSED time - from the end of backward's last
    # kernel to the completion of the last collective, i.e. how long the optimiser waited for the exchange; max over the ranks.
    comm = None
    if reducer is not None:
        try:  # (diagnostics only: `value` is already measured - a failure here must not cost the line)
            ranks = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
            dist.all_gather(ranks, torch.tensor([rank], dtype=torch.int64, device=device))
            world_seen = len({int(t.item()) for t in ranks})
            comm_steps = max(1, min(args.steps, 5))
            reducer.measure = True
            for _ in range(comm_steps):
                step()
            sync()
            reducer.measure = False
            tm = reducer.collect_timing()
            n = max(tm["steps"], 1)
            stats = torch.tensor([tm["exposed_ms"] / n, tm["allreduce_ms_sum"] / n, tm["host_wait_ms_sum"] / n], dtype=torch.float64, device=device)
            dist.all_reduce(stats, op=dist.ReduceOp.MAX)
            comm = {"world_seen": world_seen, "backend": args.dist_backend, "buckets": tm["buckets"] // n, "bytes_per_step": tm["bytes"] // n,
                    "bucket_bytes": reducer.bucket_bytes, "exposed_ms": float(stats[0]), "allreduce_ms_sum": float(stats[1]), "host_wait_ms_sum": float(stats[2]),
                    "steps": comm_steps,
                    "note": "exposed_ms = end of backward's last kernel -> last collective done (HIP events, main stream), allreduce_ms_sum = sum over the "
                            "buckets of (range final -> collective done) on the side stream, both per step, max over ranks; host_wait_ms_sum: host time "
                            "inside the collective calls (non-zero where the backend blocks the host: gloo)"}
            if args.comm_only:
                # the same buckets on their own: buffers of the sizes just seen, all-reduced in place back to back, nothing else on the GPU
                sizes = reducer.last_bucket_bytes or [tm["bytes"] // n]
                bufs = [torch.zeros(max(b // 4, 1), dtype=torch.float32, device=device) for b in sizes]
                for _ in range(3):
                    for b in bufs:
                        dist.all_reduce(b)
                sync()
                t2 = time.perf_counter()
                for _ in range(args.steps):
                    for b in bufs:
                        dist.all_reduce(b)
                sync()
                co = torch.tensor([(time.perf_counter() - t2) / args.steps * 1e3], dtype=torch.float64, device=device)
                dist.all_reduce(co, op=dist.ReduceOp.MAX)
                tot = sum(sizes)
                comm["comm_only"] = {"ms_per_step": float(co[0]), "buckets": len(sizes), "bytes": tot,
                                     "busbw_GBs": tot * 2 * (world - 1) / world / (float(co[0]) * 1e-3) / 1e9 if float(co[0]) > 0 else None,
                                     "note": "bucketed all-reduce alone, back to back on the default stream; busbw = bytes * 2 (N-1)/N / time"}
        except Exception as e:  # noqa: BLE001
            reducer.measure = False
            comm = {"error": f"{type(e).__name__}: {e}"[:300]}
    # optimiser step alone (already inside `value`; reported for reference)
    sync()
    t1 = time.perf_counter()
    for _ in range(5):
        opt.step()
    sync()
    opt_ms = (time.perf_counter() - t1) / 5 * 1e3

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        crops = args.batch * world * args.steps / elapsed
        ks = timer.summary(max(roof_steps, 1))
        if args.per_call and roof_steps:
            print("\n".join(timer.per_call(roof_steps)), file=sys.stderr)
        roof, top5 = None, []
        if ks:
            # dominant kernel = the kernel NAME (as rocprofv3 --stats lists it) with the largest total time in the
            # roofline pass.  Depthwise kernels are HBM-bound; pw_split_k runs 6 bf16 MFMA products per fp32 product,
            # so its fp32-equivalent ceiling is the dense bf16 peak / 6; pw_gemm_k / pw_wgrad_k (the early, HBM-bound
            # pointwise layers) are fp32 MFMA kernels priced against HBM.
            # counter traffic: --traffic-json, else the newest profiles/r*_pmc_traffic.json whose csrc_sha256 stamp (tools/build_id.py)
            # equals this tree's - counters of another build are never attached, and the committed summaries (tools/profile_round.sh: the
            # default workload) only to the default workload: per-launch bytes of B = 512 say nothing about another batch size or network
            traffic, traffic_src = {}, None
            default_workload = args.batch == 512 and args.backbone == "mobilenetv1" and args.precision == "fp32" and not args.blurpool
            bc_workload = args.batch == 512 and args.backbone == "mobilenetv1" and args.precision == "bf16-compute" and not args.blurpool
            cands = [args.traffic_json] if args.traffic_json else (
                sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic.json")), reverse=True) if default_workload else
                sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_traffic_bf16_compute.json")), reverse=True) if bc_workload else [])
            sys.path.insert(0, os.path.join(REPO, "tools"))
            from build_id import csrc_sha256
            tree = csrc_sha256()
            for c in cands:
                try:
                    d = json.load(open(c))
                except (OSError, ValueError):
                    continue
                if d.get("csrc_sha256") == tree or (args.traffic_json and "csrc_sha256" not in d):
                    traffic, traffic_src = d["kernels"], os.path.relpath(c, REPO)
                    break

            def roofline_of(k):
                v = ks[k]
                sec, launches = v["ms"] * 1e-3, v["calls_per_step"] * roof_steps
                products = 3 if "pw16" in k else 6 if "pw_split" in k else 0  # 16-bit MFMA products per fp32 product (the bf16-compute GEMMs, one product, are HBM-bound)
                hbm = products == 0
                if hbm:
                    ach, peak, unit = v["bytes"] / sec / 1e9, PEAK_HBM_GBS, "GB/s"
                else:
                    ach, peak, unit = v["flops"] / sec / 1e12, PEAK_BF16_MFMA_TFLOPS / products, "TFLOP/s"
                # the labels of work() are prefixes of the profiler's kernel names (trailing template arguments omitted)
                # (one label can cover several instantiations - dw_fwd_tiled_k<1, float, SKIP, 32, CARRY>: their launch-weighted mean)
                trs = [traffic[k]] if k in traffic else [v_ for k_, v_ in traffic.items() if k_.startswith(k.rstrip(">"))]
                tr_bytes = (sum(t["bytes_per_launch"] * t["launches"] for t in trs) / sum(t["launches"] for t in trs)) if trs else None
                return {"bound": "hbm" if hbm else "mfma", "kernel": k, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                        "traffic": tr_bytes, "algorithmic_bytes_per_launch": v["bytes"] / launches,
                        "flops_per_launch": v["flops"] / launches, "launch_avg_us": v["ms"] / launches * 1e3,
                        "launches_per_step": v["calls_per_step"], "ms_per_step": v["ms_per_step"]}

            order = sorted(ks, key=lambda k: -ks[k]["ms"])
            roof = roofline_of(order[0])
            roof["peak_note"] = ("HBM3E spec 8 TB/s (6.3 TB/s achievable per the MI355X guide)" if roof["bound"] == "hbm"
                                 else "fp32-equivalent: 2500 TF dense 16-bit MFMA / piece products per fp32 product (3: fp16 x 2 split, 6: bf16 x 3 split)")
            roof["traffic_source"] = (traffic_src + f" (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; csrc_sha256 {tree[:16]} = this build)") if traffic_src else None
            roof["pass"] = f"{roof_steps} extra steps after the timed region, HIP events around each conv call, single stream"
            top5 = [roofline_of(k) for k in order[:(16 if args.precision == 'bf16-compute' else 6)]]  # (bf16-compute: every depthwise instantiation)
        dominant = roof["kernel"] if roof else None
        default_run = args.batch == 512 and args.backbone == "mobilenetv1" and args.precision == "fp32" and not args.blurpool
        per_gpu = crops / world
        line = {
            "metric": f"face-crops/sec fwd+bwd @ batch {args.batch}", "value": crops, "unit": "crops/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": f"NetworkWithPointHead({args.backbone}{', --blurpool' if args.blurpool else ''}, point head on, NLL off = training-script defaults): "
                                   "zero_grad + fwd + multi-task loss + bwd" + ((" + RCCL grad all-reduce overlapped with bwd" if args.dist_backend == "nccl" else " + gloo grad all-reduce, ranks SHARING one GPU (dry run, not a result)") if world > 1 else "") + " + fused clip/Adam step",
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world, "input": "129x129x1 f32",
                       "parallelism": f"dp{world}"},
            "roofline": roof,
            "step_roofline": ({"hbm_frac_of_8TBs": per_gpu * ALGO_BYTES_PER_CROP * (0.5 if args.precision == "bf16-compute" else 1.0) / (PEAK_HBM_GBS * 1e9),
                               **({"bf16_mfma_frac_of_2500TF": per_gpu * ALGO_FLOP_PER_CROP / (PEAK_BF16_MFMA_TFLOPS * 1e12)} if args.precision == "bf16-compute"
                                  else {"fp32_mfma_frac_of_157TF": per_gpu * ALGO_FLOP_PER_CROP / (PEAK_FP32_MFMA_TFLOPS * 1e12)})} if args.backbone == "mobilenetv1"
                              else {"fp32_mfma_frac_of_157TF": per_gpu * 4.203e9 / (PEAK_FP32_MFMA_TFLOPS * 1e12)}),  # 4.203 GFLOP/crop
            "enqueue": ("hipGraph replay (1 capture)" if use_graph[0] else "eager Python launches") + enqueue_note,
            "optimizer_ms": opt_ms, "host_enqueue_ms_per_step": host_enqueue / args.steps * 1e3, "loss": float(loss.item()),
            "kernels_ms_per_step": {k: round(v["ms_per_step"], 3) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms"])},
            "gemm_family_TFLOPs": (sum(v["flops"] for k, v in ks.items() if k.startswith("pw")) /
                                   max(sum(v["ms"] for k, v in ks.items() if k.startswith("pw")) * 1e-3, 1e-12) / 1e12) if ks else None,
            "dominant_kernel": dominant, "top_kernels": top5,
        }
        if comm is not None:
            line["comm"] = comm
        if world == 1 and default_run and not args.no_legs:
            # BASELINE configs 2 and 5 on the same clock as the headline: two short legs in this process, AFTER the headline
            # fields above are final; they never change `value`
            legs = {}
            for name, (lb, lp) in (("B256", (256, "fp32")), ("bf16_compute", (512, "bf16-compute"))):
                try:
                    legs[name] = run_leg(args, device, lb, lp)
                except Exception as exc:  # noqa: BLE001
                    legs[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            legs["note"] = ("same process, after the timed region: B256 = BASELINE config 2 (fp32, batch 256), bf16_compute = config 5's bf16 leg (batch 512); "
                            "10 timed steps each after 3 warm-up steps, the whole step (optimiser inside) replayed as one hipGraph")
            line["legs"] = legs
        if world == 1 and not args.no_copy_probe:
            line["copy_probe"] = copy_probe(device)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
