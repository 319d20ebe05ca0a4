#!/usr/bin/env python3
"""Average the per-dispatch PMC values of the rollout kernel from rocprofv3 counter_collection CSVs.
usage: pmc_summary.py <commit> <dir>...   -- the summary is stamped with the commit it was taken at and with the hash of
the kernel sources of the tree it ran in (bench.py quotes it only when that hash matches the running build)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha16  # noqa: E402

commit = sys.argv[1]
out = {"_meta": {"commit": commit, "kernel_sources_sha16": kernel_sources_sha16(),
                 "command": "rocprofv3 --pmc <group> -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-pipelined-extra --no-extras",
                 "kernel_filter": "rollout_kernel"}}
for d in sys.argv[2:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "rollout" in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            out[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
print(json.dumps(out, indent=1))
