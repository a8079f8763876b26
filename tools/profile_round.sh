#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh NAME
# Collects into gpurun_out/NAME/: rocprofv3 kernel-trace + stats of bench.py, the --pmc passes (SQ counters, FETCH_SIZE,
# WRITE_SIZE — one pass each, counters only) over tools/quick_time.py, and the bench line of an un-profiled run.
set -e
name=$1
root=$(pwd)
out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" --no-cpu-baseline > "$out/bench_profiled.log" 2>&1
echo "trace done"
for wl in C3:50 C2:500; do
  w=${wl%%:*}; spp=${wl##*:}
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d "$out/${w}_sq" -- python3 "$root/tools/quick_time.py" $w $spp > "$out/${w}_sq.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/${w}_fetch" -- python3 "$root/tools/quick_time.py" $w $spp > "$out/${w}_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/${w}_write" -- python3 "$root/tools/quick_time.py" $w $spp > "$out/${w}_write.log" 2>&1
  echo "pmc $w done"
done
cd "$root"
python3 tools/pmc_summary.py "$out"/C3_sq "$out"/C3_fetch "$out"/C3_write "$out"/C2_sq "$out"/C2_fetch "$out"/C2_write > "$out/pmc_counters.txt"
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1); cp "$f" "$out/kernel_stats.csv"
f=$(find "$out/trace" -name "*kernel_trace.csv" | head -1); python3 tools/summarize_trace.py "$f" > "$out/render_kernel_dispatches.txt"
python3 bench.py > "$out/bench_line.json" 2> "$out/bench_stderr.log"
cat "$out/bench_line.json"
# the raw traces are large: keep the summaries only
rm -rf "$out/trace" "$out"/C?_sq "$out"/C?_fetch "$out"/C?_write
