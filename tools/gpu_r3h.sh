#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "tok_attention" 2>&1 | tail -3 > $O/h_ops.log
timeout 300 python tools/tokattn_probe.py timed 1 > $O/h_timed.log 2>&1
timeout 300 python tools/tokattn_probe.py timed 2 >> $O/h_timed.log 2>&1
timeout 300 python tools/tokattn_probe.py 2 > $O/h_probe.log 2>&1
for f in h_ops h_timed h_probe; do echo "== $f"; grep -v amdgpu.ids $O/$f.log | tail -12; done
