#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage remarks: kernel, VGPRs, AGPRs, scratch, occupancy, LDS."""
import re, subprocess, sys
for path in sys.argv[1:]:
    cur, rows = None, []
    for line in open(path, errors="replace"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
        for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and cur is not None:
                cur[key] = m.group(1)
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(rows, names):
        print(f"{n[:100]:100s} V={r.get('VGPRs')} A={r.get('AGPRs')} scr={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')}")
