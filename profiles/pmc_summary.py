#!/usr/bin/env python3
"""Summarises one rocprofv3 --pmc counter over the dispatches of one kernel.
usage: pmc_summary.py COUNTER_COLLECTION.csv KERNEL_SUBSTRING  -> prints JSON"""
import csv
import json
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
out = {}
for r in rows:
    out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(json.dumps({k: {"dispatches": len(v), "mean": sum(v) / len(v), "sum": sum(v), "min": min(v), "max": max(v)}
                  for k, v in out.items()}))
