#!/usr/bin/env python3
"""Print the start/end timeline of the rollout kernel from a rocprofv3 kernel trace (diagnostic)."""
import csv
import glob
import sys

for d in sys.argv[1:]:
    f = d if d.endswith(".csv") else glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "rollout" in r["Kernel_Name"] or "select" in r["Kernel_Name"] or "wait_rolled" in r["Kernel_Name"] or "ccl" in r["Kernel_Name"].lower()]
    rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))[-40:]
    t0 = int(rows[0]["Start_Timestamp"])
    print(d)
    for r in rows[:16]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"  {r['Kernel_Name'][:36]:38s} queue", r.get("Queue_Id"), f"start {s/1e3:9.2f} us  end {e/1e3:9.2f} us  dur {(e-s)/1e3:7.2f} us")
