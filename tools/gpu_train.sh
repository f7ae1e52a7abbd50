#!/bin/bash
# Training path visit: backward parity tests, fused attention backward probe at the ViT shape, train-step probe + its kernel breakdown
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_backward.py -q -x 2>&1 | tail -15 > $O/v11_pytest.log
timeout 300 python tools/flash_bwd_probe.py > $O/v11_flash_bwd.log 2>&1
timeout 600 python tools/train_step_probe.py > $O/v11_train_step.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof_train
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o train -- python $R/tools/train_step_probe.py > $O/v11_rocprof.log 2>&1
find $O/prof_train -name "*kernel_trace*" -size +8M -delete 2>/dev/null
cd $R
cat $O/v11_pytest.log; grep -v amdgpu $O/v11_flash_bwd.log; grep -v amdgpu $O/v11_train_step.log | tail -3
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_train/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
        print(f"{r['Name'][:80]:80s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
