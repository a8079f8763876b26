#!/usr/bin/env python3
"""Extracts the render-kernel dispatches from a rocprofv3 kernel-trace CSV:  summarize_trace.py TRACE.csv > out.txt"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "render_kernel" in r["Kernel_Name"]]
print("# kernel dur_ms LDS_bytes VGPRs AccumVGPRs SGPRs workgroup grid")
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print(name, "%.3f" % dur, r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Workgroup_Size_X"], r["Grid_Size_X"])
