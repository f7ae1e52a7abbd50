#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "tok_attention" 2>&1 | tail -4 > $O/i_ops.log
timeout 300 python tools/tokattn_probe.py timed > $O/i_timed.log 2>&1
timeout 900 python -m pytest tests/test_gpu_backward.py -q -k "linvt or fixture or zero1 or context or training_step or dpo" 2>&1 | tail -25 > $O/i_bwd.log
timeout 600 python -m pytest tests/test_checkpoint.py tests/test_gpu_path.py -m gpu -q 2>&1 | tail -12 > $O/i_path.log
for f in i_ops i_timed i_bwd i_path; do echo "== $f"; grep -v amdgpu.ids $O/$f.log | tail -25; done
