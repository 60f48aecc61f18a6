"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as the MI355X guide
prescribes) -> JSON {kernel: {launches, fetch_bytes, write_bytes, bytes_per_launch}}.

Corrections (MI355X_MICROARCH.md, HBM section): both counters are in KB; on gfx950 FETCH_SIZE tallies the 128-byte
requests of wide (16 B/lane) streaming reads at 64 B, so it is doubled; WRITE_SIZE is exact for 16-B/lane stores and
float atomics.

An optional third pass (TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum) adds, per kernel, the L2 hit rate
TCC_HIT / (TCC_HIT + TCC_MISS) and the read requests the CUs' L1s sent to L2 (what a GEMM tile re-streams from L2, which
the memory-side FETCH_SIZE does not see).

usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [<l2_counter_collection.csv>]"""
import collections, csv, json, re, sys


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("ttk::", "").replace("bc::", "").strip()


def collect(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = agg[short(r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
l2 = {c: collect(sys.argv[4], c) for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum")} if len(sys.argv) > 4 else None
out = {}
for k in sorted(set(fetch) | set(write)):
    if not (k.startswith("pw") or k.startswith("dw_") or k.startswith("stem") or k.startswith("bn_") or k.startswith("heads")
            or k.startswith("loss") or k.startswith("avgpool") or k.startswith("clip_adam") or k.startswith("affine") or k.startswith("bc_")
            or k.startswith("fold_")):
        continue
    nf, f = fetch.get(k, [0, 0.0])
    nw, w = write.get(k, [0, 0.0])
    n = max(nf, nw, 1)
    fb, wb = 2.0 * 1024.0 * f / max(nf, 1), 1024.0 * w / max(nw, 1)
    out[k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "bytes_per_launch": fb + wb}
    if l2 is not None and k in l2["TCC_HIT_sum"]:
        nh, hit = l2["TCC_HIT_sum"][k]
        _, miss = l2["TCC_MISS_sum"].get(k, [0, 0.0])
        _, rd = l2["TCP_TCC_READ_REQ_sum"].get(k, [0, 0.0])
        out[k].update(l2_hit_rate=hit / max(hit + miss, 1.0), l2_hits_per_launch=hit / max(nh, 1), l2_misses_per_launch=miss / max(nh, 1),
                      l1_read_requests_to_l2_per_launch=rd / max(nh, 1))
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from build_id import csrc_sha256  # noqa: E402

json.dump({"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `bench.py --steps 2 --warmup 1 --no-graph`, "
                     "KB -> bytes, FETCH_SIZE doubled (gfx950 wide-read rule)",
           "csrc_sha256": csrc_sha256(),  # the kernel sources these counters belong to (tools/build_id.py); bench.py checks it
           "kernels": out}, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["bytes_per_launch"] * kv[1]["launches"])[:12]:
    print(f"{k:40s} n={v['launches']:4d}  {v['bytes_per_launch']/1e6:9.1f} MB/launch")
