#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_backward.py -q -x -s -k "dpo" 2>&1 | tail -15 | tee $O/t_tests.log
