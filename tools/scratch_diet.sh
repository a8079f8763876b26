#!/bin/bash
# usage (GPU box, repo root): tools/scratch_diet.sh NAME — the C2 frame (1080p, 500 spp) with the pooled (pixel, sample) hand-out + per-sample scratch + ordered sum
# (default) against the unsplit launch (RMD_SAMPLE_SPLIT=1: lane = pixel, the sum stays in registers, no scratch at all): kernel time and
# L2<->fabric traffic (FETCH_SIZE / WRITE_SIZE, one pass each).  Both are bit-exact.
set -e
name=$1
root=$(pwd); out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for split in 0 1; do
  RMD_SAMPLE_SPLIT=$split python3 "$root/tools/quick_time.py" C2 500 > "$out/split$split.log" 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    RMD_SAMPLE_SPLIT=$split timeout -k 5 200 rocprofv3 --pmc $c --output-format csv -d "$out/C2-split${split}_500_$c" -- python3 "$root/tools/quick_time.py" C2 500 2 > "$out/split${split}_$c.log" 2>&1
  done
  echo "split $split done" >> "$out/progress.txt"
done
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for split in (0, 1):
    t = [l for l in open(os.path.join(out, "split%d.log" % split)) if "kernel" in l]
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob(os.path.join(out, "C2-split%d_500_%s" % (split, c), "**", "*counter_collection.csv"), recursive=True):
            v += [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c]
        vals[c] = sum(v) / max(1, len(v))
    print("RMD_SAMPLE_SPLIT=%d: %s | FETCH_SIZE %.4g KiB, WRITE_SIZE %.4g KiB -> %.2f GB per frame (2 x fetch + write)" % (
        split, t[-1].strip() if t else "?", vals["FETCH_SIZE"], vals["WRITE_SIZE"], (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024 / 1e9))
PY
rm -rf "$out"/C2-split*
