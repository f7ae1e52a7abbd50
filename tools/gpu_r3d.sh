#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "tok_attention" 2>&1 | tail -5 > $O/d_ops.log
timeout 300 python tools/tokattn_probe.py 1 > $O/d_probe.log 2>&1
timeout 300 python -m pytest tests/test_checkpoint.py -m gpu -q 2>&1 | tail -15 > $O/d_ck.log
for f in d_ops d_probe d_ck; do echo "== $f"; grep -v amdgpu.ids $O/$f.log | tail -12; done
