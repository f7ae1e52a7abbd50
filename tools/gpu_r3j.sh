#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_prefill.py -q 2>&1 | tail -30 > $O/j_prefill.log
timeout 600 python tools/prefill_probe.py > $O/j_probe.log 2>&1
timeout 300 python -m pytest tests/test_gpu_backward.py -q -k "fixture" 2>&1 | tail -3 > $O/j_fix.log
for f in j_prefill j_probe j_fix; do echo "== $f"; grep -v "amdgpu.ids\|Warning\|warn" $O/$f.log | tail -30; done
