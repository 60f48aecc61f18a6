"""stdin: bench.py output; prints the per-kernel milliseconds per step of its JSON line."""
import json, sys
for l in sys.stdin:
    if l.startswith("{"):
        d = json.loads(l)
        print(round(d["value"]), round(d["ms_per_step"], 3), d.get("loss"))
        for k, v in d["kernels_ms_per_step"].items():
            print(f"  {v:7.3f}  {k}")
