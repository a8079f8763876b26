#!/bin/bash
# usage (GPU box, repo root): tools/sq_pass.sh NAME WORKLOAD SPP — one rocprofv3 SQ-counter pass over tools/quick_time.py; prints the derived figures
set -e
name=$1; wl=$2; spp=$3
root=$(pwd); out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d "$out/${wl}_${spp}_sq" -- python3 "$root/tools/quick_time.py" $wl $spp > "$out/${wl}_sq.log" 2>&1
cd "$root"
python3 tools/pmc_summary.py --json "$out" --tag "$name" "$out/${wl}_${spp}_sq" > "$out/${wl}_counters.txt"
cat "$out/${wl}_sq.log" | grep Msamples | tail -1
python3 -c "
import json; d=json.load(open('$out/pmc_latest.json'))
for k,v in d.items(): print(k, 'wave-instr/sample %.2f  lane util %.3f  valu active/wave %.3f' % (v['wave_instr_valu_per_sample'], v['lane_utilisation'], v['valu_active_per_wave']))"
rm -rf "$out/${wl}_${spp}_sq"
