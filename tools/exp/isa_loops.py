#!/usr/bin/env python3
"""Per kernel of a hipcc -save-temps .s file: instruction mix of the whole body and of every long loop (MFMA, vector ALU, LDS, scratch, accvgpr moves, waits).
   python tools/exp/isa_loops.py file.s [name-substring]"""
import re
import sys

src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
ends = [i for i, l in enumerate(src) if l.startswith(".Lfunc_end")]


def mix(seg):
    c = lambda pat: sum(bool(re.search(pat, x)) for x in seg)
    return dict(n=len(seg), mfma=c(r"\bv_mfma"), valu=c(r"^\s+v_(?!mfma|accvgpr)"), accr=c("v_accvgpr_read"), accw=c("v_accvgpr_write"), dsr=c(r"\bds_read"), dsw=c(r"\bds_write"),
                scr_ld=c("scratch_load"), scr_st=c("scratch_store"), gld=c(r"\bglobal_load"), gst=c(r"\bglobal_store"), wait=c("s_waitcnt"), nop=c(r"\bs_nop"), salu=c(r"^\s+s_(?!waitcnt|nop|barrier)"), bar=c("s_barrier"))


for (i, name) in starts:
    if want not in name:
        continue
    e = min(x for x in ends if x > i)
    body = src[i:e]
    print(name)
    print("   body ", mix(body))
    labels = {m.group(1): k for k, l in enumerate(body) for m in [re.match(r"(\.LBB\d+_\d+):", l)] if m}
    for k, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < k and k - labels[m.group(1)] > 300:
            print("   loop ", m.group(1), mix(body[labels[m.group(1)]:k]))
