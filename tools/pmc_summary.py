#!/usr/bin/env python3
"""Average the per-dispatch PMC values of the rollout kernel from rocprofv3 counter_collection CSVs."""
import collections
import csv
import glob
import json
import sys

out = {}
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "rollout" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            out[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
print(json.dumps(out, indent=1))
