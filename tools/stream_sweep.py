"""Where does the ~5 TB/s cap of the step's HBM-bound kernels come from?  Sweeps ttk_stream_probe (csrc/probe.hip) over the
access shapes those kernels use and prints GB/s (bytes read + written) per configuration.

    python tools/stream_sweep.py [out.json]

Axes: mix (read only, 1:1 copy, 2:1, 5:1 read:write = the depthwise backward), row bytes (= 4 C of a channels-last tensor),
segment bytes (the channel slab a workgroup owns; == row bytes: a linear sweep), 16-byte loads in flight per lane,
workgroups per CU, cache policy of the loads, and the alignment of the buffers (2 MiB vs 2 MiB + 4 KiB + 128 B)."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (REPO, os.path.join(REPO, "neuralnet-tracker-traincode_amd")):
    sys.path.insert(0, p_)
import torch  # noqa: E402

import trackertraincode._hip as H  # noqa: E402

L, p = H.lib(), H.ptr
dev = torch.device("cuda", 0)
cus = torch.cuda.get_device_properties(dev).multi_processor_count
GIB = 1 << 30
pool = torch.empty(3 * GIB // 4 + (4 << 20), dtype=torch.float32, device=dev).normal_()  # 3 GiB + 16 MiB of slack for the alignment offsets
sink = torch.zeros(4, dtype=torch.float32, device=dev)


def aligned(offset_floats, n):
    base = pool.data_ptr()
    pad = (-base) % (2 << 20)  # to a 2 MiB boundary
    lo = pad // 4 + offset_floats
    return pool[lo:lo + n]


def run(nread, nwrite, row_bytes, seg, unroll, bpc, nt, misalign=0, total=GIB):
    per = (total // 4 // nread) // (row_bytes // 4) * (row_bytes // 4)
    rows = per * 4 // row_bytes
    src = aligned(misalign, per * nread)
    dst = aligned(GIB // 4 * 2 + (2 << 20) // 4 + misalign, per * max(nwrite, 1)) if nwrite else None
    def launch():
        L.call("ttk_stream_probe", p(src), p(dst), p(sink), rows, row_bytes, seg, nread, nwrite, per * 4, unroll, nt, cus * bpc)
    launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        launch()
    e1.record()
    e1.synchronize()
    return 3 * (nread + nwrite) * per * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9


results = []
def rec(tag, **kw):
    gbs = run(**kw)
    results.append(dict(tag=tag, GBs=round(gbs, 1), **kw))
    print(f"{tag:34s} {json.dumps(kw):150s} {gbs:8.1f} GB/s", flush=True)


MIXES = [("read", 1, 0), ("copy", 1, 1), ("2to1", 2, 1), ("5to1", 5, 1)]
# 1. linear sweeps: loads in flight, workgroups per CU, cache policy
for name, nr, nw in MIXES:
    for unroll in (1, 2, 4, 8):
        if unroll * nr > 40:
            continue
        for bpc in (2, 4, 8):
            for nt in (0, 1):
                rec(f"linear {name}", nread=nr, nwrite=nw, row_bytes=4096, seg=4096, unroll=unroll, bpc=bpc, nt=nt)
# 2. slab shapes: row bytes = 4 C, C = 32 ... 1024; segment = 128 / 256 / 512 bytes
for name, nr, nw in MIXES:
    for row_bytes in (128, 256, 512, 1024, 2048, 4096):
        for seg in (128, 256, 512):
            if seg > row_bytes:
                continue
            for unroll, bpc in ((4, 4), (8, 4), (4, 8)):
                if unroll * nr > 40:
                    continue
                rec(f"slab {name} C={row_bytes // 4} seg={seg}", nread=nr, nwrite=nw, row_bytes=row_bytes, seg=seg, unroll=unroll, bpc=bpc, nt=1)
# 3. alignment of the buffers (2 MiB aligned vs off by 4 KiB + 128 B)
for name, nr, nw in MIXES:
    for mis in (0, (4096 + 128) // 4):
        rec(f"align {name} +{mis * 4} B", nread=nr, nwrite=nw, row_bytes=4096, seg=4096, unroll=4 if nr < 5 else 4, bpc=4, nt=1, misalign=mis)
        rec(f"align slab {name} +{mis * 4} B", nread=nr, nwrite=nw, row_bytes=2048, seg=128, unroll=4, bpc=4, nt=1, misalign=mis)
# 4. footprint: does the rate depend on the sweep's size (Infinity Cache 256 MiB)?
for total in (GIB // 8, GIB // 2, GIB):
    rec(f"size read {total >> 20} MiB", nread=1, nwrite=0, row_bytes=4096, seg=4096, unroll=4, bpc=4, nt=1, total=total)
    rec(f"size 5to1 {total >> 20} MiB", nread=5, nwrite=1, row_bytes=4096, seg=4096, unroll=4, bpc=4, nt=1, total=total)
if len(sys.argv) > 1:
    json.dump(results, open(sys.argv[1], "w"), indent=0)
best = {}
for r in results:
    k = r["tag"]
    if k not in best or r["GBs"] > best[k]["GBs"]:
        best[k] = r
print("\n=== best per tag")
for k, r in best.items():
    print(f"{k:34s} {r['GBs']:8.1f} GB/s  unroll {r['unroll']} bpc {r['bpc']} nt {r['nt']}")
