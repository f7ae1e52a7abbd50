"""Post-process tools/gpu_pmc2.sh: per run, per kernel of interest, average counters per dispatch and derived figures.
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)   (MI355X_MICROARCH.md: the counter adds
32 cycles per v_mfma_f32_32x32x16_bf16 per SIMD; GRBM_GUI_ACTIVE = cycles the dispatch kept the GPU busy)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
KEYS = ("tok_attn2_kernel", "tok_attn_kernel", "tok_attn_combine", "flash_dp2_kernel", "flash_dp_kernel", "flash_d64_kernel", "flash_bwd_dq_kernel", "flash_bwd_dkv_kernel", "gemm_bt_drain_kernel", "gemm_skinny64_kernel", "gemm_bt_kernel", "gemm_bf16_nt_kernel<64", "gemm_bf16_nt_kernel<128, 128, true", "gemm_bf16_nt_kernel<128", "gemm_splitk_reduce")
res = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sorted(glob.glob(root + "/*_p*")):
    if not os.path.isdir(d):
        continue
    run = os.path.basename(d).rsplit("_p", 1)[0]
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((x for x in KEYS if x in r.get("Kernel_Name", "")), None)
            if k is None:
                continue
            c = res[(run, k)][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"])
            c[1] += 1
# dispatch durations from the kernel traces of the same runs (ns)
dur = collections.defaultdict(lambda: [0.0, 0])
for d in sorted(glob.glob(root + "/*_p*")):
    if not os.path.isdir(d):
        continue
    run = os.path.basename(d).rsplit("_p", 1)[0]
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((x for x in KEYS if x in r.get("Kernel_Name", "")), None)
            if k is not None:
                dur[(run, k)][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                dur[(run, k)][1] += 1
out = {}
for (run, k), ctr in res.items():
    avg = {n: v[0] / max(v[1], 1) for n, v in ctr.items()}
    e = {"kernel": k, "dispatches": max(v[1] for v in ctr.values()), **{n: round(x, 1) for n, x in avg.items()}}
    if dur[(run, k)][1]:
        e["avg_dispatch_us_under_profiler"] = round(dur[(run, k)][0] / dur[(run, k)][1] / 1e3, 1)
    if "GRBM_GUI_ACTIVE" in avg and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        e["mfma_util"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["GRBM_GUI_ACTIVE"] * 1024), 4)
    elif "SQ_VALU_MFMA_BUSY_CYCLES" in avg and dur[(run, k)][1]:
        # no GRBM_GUI_ACTIVE (that pass aborted in rocprofv3): matrix-pipe busy cycles per SIMD over the dispatch time
        # at the 2.4 GHz the 2.5 PF/s peak is quoted at (a lower bound of the busy fraction: the chip clocks lower)
        cyc = dur[(run, k)][0] / dur[(run, k)][1] * 2.4
        e["mfma_busy_frac_at_2p4GHz"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024), 4)
    if "SQ_WAVE_CYCLES" in avg:
        for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if n in avg:
                e[n.lower() + "_frac_of_wave_cycles"] = round(avg[n] / avg["SQ_WAVE_CYCLES"], 4)
    if "FETCH_SIZE" in avg:
        e["hbm_bytes_per_dispatch"] = round((2 * avg["FETCH_SIZE"] + avg.get("WRITE_SIZE", 0.0)) * 1024)
    out[f"{run}:{k}"] = e
print(json.dumps(out, indent=1))
