#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
NCCL_DEBUG=VERSION timeout 600 python -m pytest tests/test_gpu_replicas.py -q -x -s 2>&1 | tail -15 | tee $O/y_tests.log
