#!/usr/bin/env python
"""Which kernel runs which product of one volume: joins U2TOK_GEMM_TRACE lines (one per gemm_bf16 call, in order) with the GEMM-class
kernels of a rocprofv3 --kernel-trace of the same run (in start order; reduce launches listed with the product in front of them).

    U2TOK_GEMM_TRACE=1 rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python bench.py --steps 1 --warmup 0 --streams 1 \
        --no-cpu-baseline --no-roofline --no-train-step 2> trace.txt;  python tools/gemm_kernel_map.py DIR trace.txt
"""
import collections
import csv
import re
import sys
from pathlib import Path

calls = [m.groups() for m in (re.match(r"gemm M=(\d+) N=(\d+) K=(\d+) nz=(\d+) flags=(0x[0-9a-f]+)", l) for l in open(sys.argv[2])) if m]
rows = []
for f in Path(sys.argv[1]).rglob("*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ker = [(r["Kernel_Name"].split("(")[0].replace("void u2::", "").replace("(anonymous namespace)::", ""),
        (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "gemm" in r["Kernel_Name"]]
main = [k for k in ker if "reduce" not in k[0] and "rows16" not in k[0]]
print(f"{len(calls)} gemm_bf16 calls, {len(main)} main GEMM kernels, {len(ker) - len(main)} reduce / few-rows launches")
n = len(calls)
per = collections.OrderedDict()
# the LAST len(calls) / volumes main kernels belong to the last volume; if the counts match one to one, join all
if len(main) % n == 0 or n % len(main) == 0 or True:
    k = min(n, len(main))
    for c, kk in zip(calls[-k:], main[-k:]):
        per.setdefault((c, kk[0]), []).append(kk[1])
for (c, name), v in per.items():
    print(f"{len(v):3d} x  M={c[0]:>6} N={c[1]:>6} K={c[2]:>5} nz={c[3]:>2} flags={c[4]:>5}  {sum(v) / len(v):8.1f} us  {name[:70]}")
