#!/bin/bash
# usage (GPU box, repo root): tools/stall_pass.sh NAME WORKLOAD SPP — where a wave's cycles go (SQ counters, one pass)
set -e
name=$1; wl=$2; spp=$3
root=$(pwd); out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d "$out/${wl}_${spp}_stall" -- python3 "$root/tools/quick_time.py" $wl $spp > "$out/${wl}_stall.log" 2>&1
cd "$root"
python3 tools/pmc_summary.py "$out/${wl}_${spp}_stall" | grep -v sum_kernel | tee "$out/${wl}_stall_counters.txt"
rm -rf "$out/${wl}_${spp}_stall"
