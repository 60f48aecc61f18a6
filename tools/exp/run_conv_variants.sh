#!/bin/bash
# runs tools/bench_conv.py (weight-gradient column) with the default library and every variant in tools/exp/_build (gpurun side)
cd "$(dirname "$0")/../.."
F="${FILTER:-l1 |l2 |l3 |l4 |totals}"
echo "== default"; python tools/bench_conv.py 512 3 2>/dev/null | grep -E "$F" | sed "s/| fwd.*| wgrad/| wgrad/"
for lib in tools/exp/_build/libttk_*.so; do echo "== $lib"; TTK_LIB=$PWD/$lib python tools/bench_conv.py 512 3 2>/dev/null | grep -E "$F" | sed "s/| fwd.*| wgrad/| wgrad/"; done
