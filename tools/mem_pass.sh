#!/bin/bash
# usage (GPU box, repo root): tools/mem_pass.sh NAME WORKLOAD SPP — memory-pipeline counters of the render kernel (rocprofv3 --pmc, counters only, one
# pass per group): L2 hit rate, L1 accesses / latency, TA / TCP / TD busy and stall cycles, the SQ's wait and VMEM / LDS figures.
set -e
name=$1; wl=$2; spp=$3
root=$(pwd); out=$root/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "TCP_TOTAL_READ_sum TCP_TOTAL_ACCESSES_sum" \
            "TA_TA_BUSY_sum TD_TD_BUSY_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
            "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
            "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout -k 5 90 rocprofv3 --pmc $ctrs --output-format csv -d "$out/pass$i" -- python3 "$root/tools/quick_time.py" $wl $spp 2 > "$out/pass$i.log" 2>&1 || echo "pass $i ($ctrs) failed"
  echo "pass $i done" >> "$out/progress.txt"
done
cd "$root"
python3 - "$out" <<'PY'
import collections, csv, glob, os, sys
out = sys.argv[1]
tot = {}
for f in sorted(glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in per.items():
        tot[k] = sum(v) / len(v)
for k in sorted(tot):
    print("%-40s %.6g" % (k, tot[k]))
g = lambda k: tot.get(k, float("nan"))
print("L2 hit rate %.3f" % (g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
print("L1 hit rate (1 - TCC read req / L1 cache accesses) %.3f" % (1 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum")))
print("mean L1->L2 read latency, cycles %.0f" % (g("TCP_TCC_READ_REQ_LATENCY_sum") / g("TCP_TCC_READ_REQ_sum")))
print("mean L1 latency, cycles %.0f" % (g("TCP_TCP_LATENCY_sum") / g("TCP_TOTAL_ACCESSES_sum")))
PY
rm -rf "$out"/pass? "$out"/pass??
