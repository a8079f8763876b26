#!/bin/bash
# usage (GPU box, repo root): tools/mix_pass.sh NAME WORKLOAD SPP — the render kernel's vector-instruction mix (rocprofv3 --pmc, counters only): f64 add / mul / fma /
# transcendental, 32- and 64-bit integer, conversions; with the measured issue cost of each class (profiles/r03_valu_rate.txt) the time the
# vector ALUs need for the launch's instruction stream, to hold against its duration.
set -e
name=$1; wl=$2; spp=$3
root=$(pwd); out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
            "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 "$root/tools/quick_time.py" $wl $spp 2 > "$out/pass$i.log" 2>&1 || echo "pass $i ($ctrs) failed"
  echo "pass $i done" >> "$out/progress.txt"
done
cd "$root"
python3 - "$out" "$wl" "$spp" <<'PY'
import collections, csv, glob, os, sys
out, wl, spp = sys.argv[1], sys.argv[2], int(sys.argv[3])
tot = {}
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        tot[k] = sum(v) / len(v)
n = (3840 * 2160 if wl.startswith("C4") else 1920 * 1080) * spp
for k in sorted(tot):
    print("%-28s %.6g   per sample %.3f" % (k, tot[k], tot[k] / n))
g = lambda k: tot.get(k, 0.0)
f64 = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64")
trans = g("SQ_INSTS_VALU_TRANS_F64")
other = g("SQ_INSTS_VALU") - f64 - trans
# ns per wave instruction per SIMD: f64 arithmetic 2.05, rcp / rsq 6.7, everything else (32-bit integer, moves, compares, selects) ~1.0 .. 1.3
for c_other in (0.95, 1.28):
    t = (f64 * 2.05 + trans * 6.7 + other * c_other) / 1024.0 * 1e-6
    print("VALU time of the instruction stream with 'other' at %.2f ns: %.1f ms (f64 %.1f + trans %.1f + other %.1f)" % (c_other, t, f64 * 2.05 / 1024e6, trans * 6.7 / 1024e6, other * c_other / 1024e6))
print(open(os.path.join(out, "pass1.log")).read().strip().split("\n")[-2])
PY
rm -rf "$out"/pass?
