#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" 2>&1 | tail -8 > $O/o_tests.log
tail -4 $O/o_tests.log
timeout 600 python tools/bt_sweep.py 2>&1 | grep -v Warn | tee $O/o_sweep.log
