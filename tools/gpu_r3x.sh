#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_backward.py -q -x -k "rccl or adamw" 2>&1 | tail -15 | tee $O/x_tests.log
