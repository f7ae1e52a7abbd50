#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x 2>&1 | tail -5 > $O/g_ops.log
timeout 600 python -m pytest tests/test_gpu_path.py -q -x 2>&1 | tail -5 > $O/g_path.log
timeout 600 python tools/ab_bench.py base: noflash:tok_flash=0 base1::1 noflash1:tok_flash=0:1 > $O/g_ab.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_g
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g -o k -- python $R/bench.py --steps 6 --warmup 1 --repeats 1 --streams 1 --no-cpu-baseline --no-roofline --no-train-step > $O/g_rocprof.log 2>&1
find $O/prof_g -name "*kernel_trace*" -delete 2>/dev/null
cd $R
for f in g_ops g_path g_ab; do echo "== $f"; grep -v amdgpu.ids $O/$f.log | tail -8; done
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_g/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "u2::" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"== u2 kernel time per volume {tot/7e6:.3f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:18]:
    print(f"  {r['Name'][:70]:70s} {int(r['Calls'])/7:7.1f}/vol {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/7e6:7.3f} ms/vol")
PY
