#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_k
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_k -o k -- python $R/tools/prefill_prof.py > $O/k_rocprof.log 2>&1
find $O/prof_k -name "*kernel_trace*" -delete 2>/dev/null
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_k/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows if "u2::" in r["Name"] or "Cijk" in r["Name"] or "at::" in r["Name"])
print(f"kernel time per prefill {tot/5e6:.3f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print(f"  {r['Name'][:84]:84s} {int(r['Calls'])/5:7.1f}/prefill {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/5e6:7.3f} ms")
PY
