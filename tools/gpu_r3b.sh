#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "tok_attention" 2>&1 | tail -5 > $O/b_ops.log
timeout 600 python tools/tokattn_probe.py > $O/b_probe.log 2>&1
timeout 300 python -m pytest tests/test_checkpoint.py -m gpu -q 2>&1 | tail -15 > $O/b_ck.log
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  rm -rf $O/prof_b$v
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b$v -o k -- python $R/bench.py --steps 6 --warmup 1 --repeats 1 --streams 1 --no-cpu-baseline --no-roofline --no-train-step --option tok_flash=$v > $O/b_rocprof$v.log 2>&1
  find $O/prof_b$v -name "*kernel_trace*" -delete 2>/dev/null
done
cd $R
for f in b_ops b_probe b_ck; do echo "== $f"; grep -v amdgpu.ids $O/$f.log | tail -30; done
python - <<'PY'
import csv, glob
for v in (1, 0):
    f = glob.glob(f"gpurun_out/prof_b{v}/**/*kernel_stats.csv", recursive=True)
    if not f: print("no stats", v); continue
    rows = [r for r in csv.DictReader(open(f[0])) if "u2::" in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"== tok_flash={v}: u2 kernel time per volume {tot/7e6:.3f} ms")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
        print(f"  {r['Name'][:70]:70s} {int(r['Calls'])/7:7.1f}/vol {float(r['AverageNs'])/1e3:8.1f} us {float(r['TotalDurationNs'])/7e6:7.3f} ms/vol")
PY
