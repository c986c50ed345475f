#!/usr/bin/env python3
"""The clock the chip holds under a search, from one rocprofv3 --pmc GRBM_GUI_ACTIVE pass
over profiles/query_data_dependence.py: per dispatch of the scoring kernel, busy cycles
(the counter is summed over the 8 XCDs) / duration.  The script under the profiler runs
four phases of 69 searches (random, zero, random, one-hot queries) in that order.
usage: clock_under_load.py COUNTER_COLLECTION.csv KERNEL_SUBSTRING -> JSON"""
import csv
import json
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["random", "zeros", "random (again)", "one-hot"]
per = len(rows) // 4
out = {"kernel": sys.argv[2], "dispatches": len(rows), "phases": []}
for p in range(4):
    part = rows[p * per:(p + 1) * per][3:]          # the first searches of a phase warm up
    cyc = [float(r["Counter_Value"]) / 8.0 for r in part]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in part]
    out["phases"].append({"queries": names[p], "dispatches": len(part), "mean_ms": sum(dur) / len(dur) / 1e6,
                          "mean_cycles": sum(cyc) / len(cyc), "clock_GHz": sum(cyc) / sum(dur)})
print(json.dumps(out))
