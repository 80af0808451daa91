#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_summary.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "k_render"
agg = collections.defaultdict(float)
n = 0
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            n += 1
for k in sorted(agg):
    print("%-28s %.6g" % (k, agg[k]))
if "SQ_THREAD_CYCLES_VALU" in agg and "SQ_ACTIVE_INST_VALU" in agg:
    print("lane utilisation             %.3f" % (agg["SQ_THREAD_CYCLES_VALU"] / (64 * agg["SQ_ACTIVE_INST_VALU"])))
