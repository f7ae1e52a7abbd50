#!/usr/bin/env python
"""Median / minimum duration of the kernels whose name contains `pattern`, grouped by (kernel, grid size), from a rocprofv3
--kernel-trace csv:   python tools/trace_kernel_groups.py <dir> <pattern>"""
import collections
import csv
import sys
from pathlib import Path

d = collections.OrderedDict()
for f in Path(sys.argv[1]).rglob("*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if sys.argv[2] not in n:
            continue
        grid = r.get("Grid_Size") or r.get("Grid_Size_X") or "?"
        d.setdefault((n.split("(")[0], grid), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    v = sorted(v)
    print(f"{k[0]:45s} grid {k[1]:>9s}  n={len(v):4d}  median {v[len(v) // 2] / 1e3:7.2f} us  min {v[0] / 1e3:7.2f}")
