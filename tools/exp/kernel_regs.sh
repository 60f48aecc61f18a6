#!/bin/bash
# Register / LDS / spill summary of every kernel of one translation unit:  bash tools/exp/kernel_regs.sh pwconv_r.hip [-D flags] [filter]
R=$(cd "$(dirname "$0")/../.." && pwd); C=$R/neuralnet-tracker-traincode_amd/csrc
src=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -I$R/include "$@" -o /tmp/kregs_$$.s $C/$src 2>/dev/null
python3 - /tmp/kregs_$$.s <<'PY'
import re, sys
t = open(sys.argv[1]).read()
for m in re.finditer(r"- \.agpr_count.*?\.wavefront_size", t, re.S):
    b = m.group(0)
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
    print(f"{g('vgpr_count'):>4} vgpr {g('vgpr_spill_count'):>3} spill {g('sgpr_count'):>4} sgpr {g('group_segment_fixed_size'):>7} lds  {g('name')}")
PY
rm -f /tmp/kregs_$$.s
