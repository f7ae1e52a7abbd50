#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_q
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_q -o q -- python $R/tools/train_prof.py > $O/q_rocprof.log 2>&1
find $O/prof_q -name "*kernel_trace*" -delete 2>/dev/null
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_q/**/*kernel_stats.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if "distribution" not in r["Name"] and "fill" not in r["Name"].lower()]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step {tot/4e6:.3f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print(f"  {r['Name'][:96]:96s} {int(r['Calls'])/4:7.1f}/step {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/4e6:7.3f} ms")
PY
tail -2 $O/q_rocprof.log
