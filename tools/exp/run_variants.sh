#!/bin/bash
# runs tools/bench_gemm.py with the default library and with every variant library in tools/exp/_build (gpurun side)
cd "$(dirname "$0")/../.."
F="${FILTER:-dw3_1|dw4_1|dw5_x|dw6|totals}"
echo "== default"; python tools/bench_gemm.py 512 2>/dev/null | grep -E "$F"
for lib in tools/exp/_build/libttk_*.so; do echo "== $lib"; TTK_LIB=$PWD/$lib python tools/bench_gemm.py 512 2>/dev/null | grep -E "$F"; done
