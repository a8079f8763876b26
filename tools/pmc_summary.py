#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per counter for the render kernel:  pmc_summary.py DIR [DIR...]
Prints counter totals per dispatch (averaged over the render_kernel dispatches of each pass)."""
import csv, glob, os, sys, collections

for d in sys.argv[1:]:
    for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            kn = r["Kernel_Name"]
            if "render_kernel" in kn:
                tag = ""
            elif "sum_kernel" in kn:
                tag = "[sum_kernel]"
            else:
                continue
            per[r["Counter_Name"] + tag].append(float(r["Counter_Value"]))
        for k, v in per.items():
            print("%-28s %-24s n=%d avg/dispatch=%.6g" % (os.path.basename(d), k, len(v), sum(v) / len(v)))
