#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh NAME
# Collects into gpurun_out/NAME/: rocprofv3 kernel-trace + stats of bench.py (C2 at 500 spp, then the roofline leg: C3 at its
# configured 500 spp, one launch per step), the --pmc passes (SQ counters, FETCH_SIZE, WRITE_SIZE — one pass each, counters only; C3 in both black-path modes, C2)
# over tools/quick_time.py at the same sizes, the JSON files bench.py reads (pmc_latest.json, hbm_traffic.json) and the bench
# line of an un-profiled run.  Copy the directory's files into profiles/NAME/ (and the two JSONs into profiles/) to commit them.
set -e
name=$1
root=$(pwd)
out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" --no-cpu-baseline --lean > "$out/bench_profiled.log" 2>&1
echo "trace done" | tee -a "$out/progress.txt"
for wl in C3:500 C3-end:500 C2:500; do
  w=${wl%%:*}; spp=${wl##*:}
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d "$out/${w}_${spp}_sq" -- python3 "$root/tools/quick_time.py" $w $spp > "$out/${w}_sq.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${w}_${spp}_fetch" -- python3 "$root/tools/quick_time.py" $w $spp > "$out/${w}_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${w}_${spp}_write" -- python3 "$root/tools/quick_time.py" $w $spp > "$out/${w}_write.log" 2>&1
  # the vector-instruction mix (two passes): what the instruction stream costs the vector ALUs at the measured issue cost of each class
  timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d "$out/${w}_${spp}_mixa" -- python3 "$root/tools/quick_time.py" $w $spp 2 > "$out/${w}_mixa.log" 2>&1
  timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d "$out/${w}_${spp}_mixb" -- python3 "$root/tools/quick_time.py" $w $spp 2 > "$out/${w}_mixb.log" 2>&1
  # where a wave's cycles go and what the LDS does (round 5: these used to stay in gpurun_out/)
  timeout -k 5 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA --output-format csv -d "$out/${w}_${spp}_wait" -- python3 "$root/tools/quick_time.py" $w $spp 2 > "$out/${w}_wait.log" 2>&1
  echo "pmc $w done" | tee -a "$out/progress.txt"
done
cd "$root"
python3 tools/pmc_summary.py --json "$out" --tag "profiles/$name" "$out"/C3_500_sq "$out"/C3_500_fetch "$out"/C3_500_write "$out"/C3-end_500_sq "$out"/C3-end_500_fetch "$out"/C3-end_500_write "$out"/C2_500_sq "$out"/C2_500_fetch "$out"/C2_500_write "$out"/C3_500_mixa "$out"/C3_500_mixb "$out"/C3-end_500_mixa "$out"/C3-end_500_mixb "$out"/C2_500_mixa "$out"/C2_500_mixb "$out"/C3_500_wait "$out"/C3-end_500_wait "$out"/C2_500_wait > "$out/pmc_counters.txt"
tools/kernel_regs.sh > "$out/kernel_regs.txt" 2>&1 || true  # registers / spills / scratch of every kernel of the library, from the compiler's metadata
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1); cp "$f" "$out/kernel_stats.csv"
f=$(find "$out/trace" -name "*kernel_trace.csv" | head -1); python3 tools/summarize_trace.py "$f" > "$out/render_kernel_dispatches.txt"
# the un-profiled bench line, reading the traffic / VALU figures just collected
cp "$out/pmc_latest.json" "$out/hbm_traffic.json" profiles/
python3 bench.py > "$out/bench_line.json" 2> "$out/bench_stderr.log"
cat "$out/bench_line.json"
# the raw traces are large: keep the summaries only
rm -rf "$out/trace" "$out"/C?*_500_sq "$out"/C?*_500_fetch "$out"/C?*_500_write "$out"/C?*_500_mix? "$out"/C?*_500_wait
