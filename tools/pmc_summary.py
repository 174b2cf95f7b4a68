#!/usr/bin/env python3
"""Average PMC counters per dispatch of a kernel from rocprofv3 counter_collection CSVs.
usage: pmc_summary.py <dir with pass*/p_counter_collection.csv> [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "rollout_kernel"
acc, cnt = defaultdict(float), defaultdict(int)
for f in sorted(glob.glob(os.path.join(root, "pass*", "*counter_collection.csv"))):
    for row in csv.DictReader(open(f)):
        if kern not in row["Kernel_Name"]:
            continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[row["Counter_Name"]] += 1
print(f"kernel ~ {kern}: per-dispatch averages")
for k in sorted(acc):
    print(f"  {k:28s} {acc[k] / cnt[k]:16.1f}   (n={cnt[k]})")
