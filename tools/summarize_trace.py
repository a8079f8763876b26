#!/usr/bin/env python3
"""Extracts the render-path dispatches from a rocprofv3 kernel-trace CSV:  summarize_trace.py TRACE.csv > out.txt
One line per dispatch of render_kernel / sum_kernel, in launch order, with its start relative to the first one and the gap
to the end of the previous dispatch — an outlier can then be told from a neighbour that overlapped it."""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "render_kernel" in r["Kernel_Name"] or "sum_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("# kernel start_ms gap_ms dur_ms LDS_bytes VGPRs AccumVGPRs SGPRs workgroup grid")
t0 = int(rows[0]["Start_Timestamp"]) if rows else 0
prev_end = t0
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(name, "%.3f" % ((s - t0) / 1e6), "%.3f" % ((s - prev_end) / 1e6), "%.3f" % ((e - s) / 1e6), r["LDS_Block_Size"], r["VGPR_Count"],
          r["Accum_VGPR_Count"], r["SGPR_Count"], r["Workgroup_Size_X"], r["Grid_Size_X"])
    prev_end = e
