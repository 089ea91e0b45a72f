#!/usr/bin/env python3
"""Development aid: mean counter values per launch of the kernels whose name contains <needle>, over every
rocprofv3 --pmc output directory given.   usage: pmc_means.py <needle> dir [dir ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict

needle = sys.argv[1]
sums, counts, micros = defaultdict(float), defaultdict(int), []
for directory in sys.argv[2:]:
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as handle:
            for row in csv.DictReader(handle):
                if needle in row["Kernel_Name"]:
                    sums[row["Counter_Name"]] += float(row["Counter_Value"])
                    counts[row["Counter_Name"]] += 1
    for path in glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True):
        with open(path, newline="") as handle:
            for row in csv.DictReader(handle):
                if needle in row["Kernel_Name"]:
                    micros.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for name in sorted(sums):
    print(f"{name:28s} {sums[name] / counts[name]:16.1f}   ({counts[name]} launches)")
if micros:
    print(f"{'kernel us under profiler':28s} {sum(micros) / len(micros):16.1f}")
