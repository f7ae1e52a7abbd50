#!/usr/bin/env python
"""Prints the kernel table of a bench.py JSON line (stdin or file): per kernel launches / volume, average us, ms per volume."""
import json
import sys
for ln in (open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin):
    if ln.startswith("{"):
        d = json.loads(ln)
        print("value", d["value"], d["unit"], "ms_per_step", d["ms_per_step"], "one_stream", d.get("value_one_stream"))
        kt = d.get("kernel_table")
        if kt:
            for k in kt["kernels"]:
                print(f'{k["launches_per_volume"]:6.1f} x {k["avg_us"]:8.2f} us = {k["ms_per_volume"]:7.4f} ms  [{k["class"]}] {k["kernel"][:90]}')
            print({k: v["ms_per_volume"] for k, v in kt["classes"].items()}, "sum", kt["sum_ms_per_volume"])
        for key in ("roofline", "roofline_attention", "roofline_tokenizer_attention"):
            if key in d:
                r = d[key]
                print(key, "frac", r["frac"], "achieved", r["achieved"], "avg_us", r["avg_launch_us"], "traffic", r["traffic"], "alg", r["algorithmic_bytes_per_launch"])
