#!/usr/bin/env python3
"""Copies a rocprofv3 *_kernel_stats.csv with kernel names cut to 120 chars
(torch's templated names run to kilobytes).  usage: trim_stats.py IN OUT"""
import csv
import sys

rows = list(csv.reader(open(sys.argv[1])))
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0][:120]
        w.writerow(r)
