#!/bin/bash
# usage: tools/kernel_regs.sh [EXTRA flags]  — compiles kernels.hip to ISA (no link) and prints registers / spills / scratch per kernel
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
out=${TMPDIR:-/tmp}/rmd_kernels.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -DRMD_DIAG=0 $1 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I/opt/rocm/include -S --cuda-device-only -o "$out" "$root/raymond_amd/csrc/kernels.hip" 2>/dev/null
python3 - "$out" <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(\S+)(.*?)\.wavefront_size", txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, body) or [0, "?"])[1]
    short = re.sub(r"_ZN3rmd13render_kernelILi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)EE.*", r"render_kernel<\1,\2,\3,\4,\5>", name)
    short = re.sub(r"_ZN3rmd(\d+)([a-z_]+)E.*", lambda m: m.group(2)[: int(m.group(1))], short)
    print("%-28s vgpr %3s spill %3s sgpr %3s scratch %4s lds %5s" % (short[:28], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
PY
