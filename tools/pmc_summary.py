#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per counter for the render kernel:  pmc_summary.py [--json OUT.json --tag NAME] DIR [DIR...]
Prints counter totals per dispatch (averaged over the render_kernel dispatches of each pass).  Directory names are
<WORKLOAD>[-end|-trace]_<spp>_<sq|fetch|write> (tools/profile_round.sh; the suffix is tools/quick_time.py's flags selector and becomes
part of the key: "C3" = flags 0, "C3_end" = RMD_RENDER_END_BLACK_PATHS).  Every entry records the hash of the sources it was measured at
(bench.py: source_hash) so that a later bench line can say when it has gone stale.  With --json, also writes what bench.py reads:
  profiles/pmc_latest.json   per workload: VALU issue occupancy, lane utilisation, wave instructions per sample
  profiles/hbm_traffic.json  per workload: L2<->fabric bytes per launch = 2 x FETCH_SIZE KiB (gfx950 reports half the bytes of
                             16-B/lane loads, MI355X_MICROARCH.md) + WRITE_SIZE KiB, render kernel + sum kernel
"""
import collections
import csv
import glob
import json
import os
import sys

args = sys.argv[1:]
json_out = tag = None
while args and args[0].startswith("--"):
    if args[0] == "--json":
        json_out = args[1]
    elif args[0] == "--tag":
        tag = args[1]
    args = args[2:]

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (source_hash only; importing bench.py runs nothing)

SRC_HASH = bench.source_hash()
SAMPLES = {"C2": 1920 * 1080, "C3": 1920 * 1080, "C5": 1920 * 1080, "C4": 3840 * 2160}
N_SIMD = 256 * 4
agg = collections.defaultdict(dict)  # (workload, spp) -> counter -> avg per dispatch
for d in args:
    base = os.path.basename(d.rstrip("/"))
    parts = base.split("_")
    key = (parts[0].replace("-", "_"), int(parts[1])) if len(parts) >= 3 and parts[1].isdigit() else (parts[0].replace("-", "_"), None)
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "render_kernel" in kn:
                t = ""
            elif "sum_kernel" in kn:
                t = "[sum_kernel]"
            else:
                continue
            per[r["Counter_Name"] + t].append(float(r["Counter_Value"]))
        for k, v in per.items():
            print("%-28s %-24s n=%d avg/dispatch=%.6g" % (base, k, len(v), sum(v) / len(v)))
            agg[key][k] = sum(v) / len(v)

if json_out:
    valu, traffic = {}, {"_how": __doc__.split("With --json")[1].strip()}
    for (wl, spp), c in agg.items():
        if "SQ_INSTS_VALU" in c and spp:
            n = SAMPLES[wl.split("_")[0]] * spp
            valu[wl] = {
                "source": tag, "spp": spp, "source_hash": SRC_HASH, "sq_insts_salu_per_valu": c.get("SQ_INSTS_SALU", 0.0) / c["SQ_INSTS_VALU"],
                "wave_instr_valu_per_sample": c["SQ_INSTS_VALU"] / n,
                "lane_utilisation": c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_ACTIVE_INST_VALU"] * 64.0),
                # fraction of a wave's resident cycles in which it has a VALU instruction in flight; times the waves resident per
                # SIMD (4 for both render kernels) = how busy the SIMD's vector ALU is
                "valu_active_per_wave": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
                "wave_instr_valu_per_launch": c["SQ_INSTS_VALU"],
            }
            if "SQ_INSTS_VALU_FMA_F64" in c and "SQ_INSTS_VALU_INT32" in c:
                # the instruction mix, per sample, and the time the vector ALUs need for this stream: ns per wave instruction per SIMD measured with
                # tools/microbench/valu_rate.hip (profiles/r03_valu_rate.txt): f64 add / mul / fma 2.05, v_rcp / v_rsq_f64 6.7, everything else
                # (32-bit integer and bit operations, moves, selects; f64 compares priced as those) 0.95 .. 1.28
                f64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"]
                trans = c["SQ_INSTS_VALU_TRANS_F64"]
                other = c["SQ_INSTS_VALU"] - f64 - trans
                valu[wl]["mix_per_sample"] = {"f64_add": c["SQ_INSTS_VALU_ADD_F64"] / n, "f64_mul": c["SQ_INSTS_VALU_MUL_F64"] / n, "f64_fma": c["SQ_INSTS_VALU_FMA_F64"] / n,
                                              "f64_rcp_rsq": trans / n, "int32": c["SQ_INSTS_VALU_INT32"] / n, "int64": c["SQ_INSTS_VALU_INT64"] / n, "cvt": c["SQ_INSTS_VALU_CVT"] / n,
                                              "moves_selects_compares": (other - c["SQ_INSTS_VALU_INT32"] - c["SQ_INSTS_VALU_INT64"] - c["SQ_INSTS_VALU_CVT"]) / n,
                                              "salu": c.get("SQ_INSTS_SALU", 0.0) / n, "branch": c.get("SQ_INSTS_BRANCH", 0.0) / n, "lds": c.get("SQ_INSTS_LDS", 0.0) / n,
                                              "vmem_read": c.get("SQ_INSTS_VMEM_RD", 0.0) / n, "smem": c.get("SQ_INSTS_SMEM", 0.0) / n}
                valu[wl]["valu_stream_ms"] = {"low": (f64 * 2.05 + trans * 6.7 + other * 0.95) / N_SIMD * 1e-6, "high": (f64 * 2.05 + trans * 6.7 + other * 1.28) / N_SIMD * 1e-6,
                                              "how": "sum over instruction classes of count x measured issue cost / 1,024 SIMDs; low / high = the other-class cost of 0.95 / 1.28 ns"}
            if "SQ_WAIT_ANY" in c and "SQ_LDS_BANK_CONFLICT" in c:
                # where a wave's resident cycles go, and the LDS: bank-conflict cycles over the cycles the LDS is busy with this kernel's instructions
                valu[wl]["wait"] = {"SQ_WAIT_ANY_per_wave_cycle": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "SQ_WAIT_INST_ANY_per_wave_cycle": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
                                    "SQ_WAIT_INST_LDS_per_wave_cycle": c.get("SQ_WAIT_INST_LDS", 0.0) / c["SQ_WAVE_CYCLES"],
                                    "LdsBankConflict_over_lds_active": c["SQ_LDS_BANK_CONFLICT"] / max(c.get("SQ_ACTIVE_INST_LDS", 0.0), 1.0),
                                    "LdsBankConflict_over_lds_idx_active": c["SQ_LDS_BANK_CONFLICT"] / max(c.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0)}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            fetch = c["FETCH_SIZE"] + c.get("FETCH_SIZE[sum_kernel]", 0.0)
            write = c["WRITE_SIZE"] + c.get("WRITE_SIZE[sum_kernel]", 0.0)
            traffic[wl] = {"source": tag, "spp": spp, "source_hash": SRC_HASH, "bytes_per_launch": 2.0 * fetch * 1024.0 + write * 1024.0,
                           "render_fetch_kib": c["FETCH_SIZE"], "render_write_kib": c["WRITE_SIZE"],
                           "sum_fetch_kib": c.get("FETCH_SIZE[sum_kernel]", 0.0), "sum_write_kib": c.get("WRITE_SIZE[sum_kernel]", 0.0)}
    os.makedirs(json_out, exist_ok=True)
    with open(os.path.join(json_out, "pmc_latest.json"), "w") as f:
        json.dump(valu, f, indent=1)
    with open(os.path.join(json_out, "hbm_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
