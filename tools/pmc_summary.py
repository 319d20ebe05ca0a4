#!/usr/bin/env python3
"""Average the per-dispatch PMC values of the rollout kernel from rocprofv3 counter_collection CSVs.
usage: pmc_summary.py [--command "<profiled command>"] [--kernel <substring>] <commit> <dir>...
The summary is stamped with the commit it was taken at and with the hash of the kernel sources of the tree it ran in
(bench.py quotes it only when that hash matches the running build)."""
import argparse
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha16  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--command", default="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-pipelined-extra --no-extras")
ap.add_argument("--kernel", default="rollout", help="substring of the kernel name whose dispatches are averaged")
ap.add_argument("--min-grid", type=int, default=0, help="only dispatches whose Grid_Size is at least this (separates the batched launch from single ones)")
ap.add_argument("commit")
ap.add_argument("dirs", nargs="+")
a = ap.parse_args()
out = {"_meta": {"commit": a.commit, "kernel_sources_sha16": kernel_sources_sha16(),
                 "command": f"rocprofv3 --pmc <group> -- {a.command}", "kernel_filter": a.kernel}}
for d in a.dirs:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if a.kernel in row["Kernel_Name"] and int(row.get("Grid_Size", "0") or 0) >= a.min_grid:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
                for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
                    if k in row and row[k] not in ("", None):
                        out["_meta"].setdefault("dispatch", {})[k] = row[k]
                out["_meta"]["kernel_name"] = row["Kernel_Name"][:160]
        for k, v in agg.items():
            out[k] = {"launches": len(v), "mean": sum(v) / len(v), "min": min(v), "max": max(v)}
print(json.dumps(out, indent=1))
