"""Post-process rocprofv3 --pmc passes of bench.py into HBM traffic per step for every kernel class.

    python tools/pmc_traffic.py <dir with */*counter_collection.csv> <steps profiled> > traffic.json

FETCH_SIZE / WRITE_SIZE are in KiB (hbm_bytes = counter * 1024).  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE
reports half of the bytes of a wide coalesced streaming read -> the read side is doubled ("read_x2"); WRITE_SIZE is
uncalibrated there and is reported as is.  Dispatches are grouped by kernel-name prefix."""
import collections
import csv
import glob
import json
import sys

root, steps = sys.argv[1], int(sys.argv[2])
cls = [("gemm_bf16_nt_kernel", "gemm_bf16"), ("gemm_bt_kernel", "gemm_bf16"), ("gemm_bt_drain_kernel", "gemm_bf16"),
       ("gemm_skinny", "gemm_bf16"), ("gemm_splitk", "gemm_bf16"), ("gemm_pp", "gemm_bf16"), ("gemm_rows16", "gemm_bf16"), ("flash_", "flash_d64"), ("tok_attn", "tok_attention"),
       ("temporal_attention", "temporal_attention_kernel"), ("layernorm", "layernorm_kernel"),
       ("softmax_rows", "softmax_rows_kernel"), ("transpose", "transpose_kernel"), ("im2col", "im2col_kernel")]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        name = next((lab for key, lab in cls if key in k), None)
        if name is None:
            continue
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[name][r["Counter_Name"]] += 1
out = {}
for name, c in agg.items():
    fetch, write = c.get("FETCH_SIZE", 0.0) * 1024, c.get("WRITE_SIZE", 0.0) * 1024
    n = max(cnt[name].get("FETCH_SIZE", 0), cnt[name].get("WRITE_SIZE", 0))
    out[name] = {"launches_per_step": n / steps, "fetch_bytes_per_step_raw": fetch / steps,
                 "read_x2_bytes_per_step": 2 * fetch / steps, "write_bytes_per_step": write / steps,
                 "hbm_bytes_per_step": (2 * fetch + write) / steps,
                 "hbm_bytes_per_launch": (2 * fetch + write) / max(n, 1)}
print(json.dumps({"steps": steps, "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of wide "
                  "coalesced reads); WRITE_SIZE as reported", "kernels": out}, indent=1))
