#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "flash" 2>&1 | tail -6
timeout 300 python tools/gpu_check.py flashtime flashperf > $O/flashtime.log 2>&1; echo "flashtime exit $?"
tail -26 $O/flashtime.log
