#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_configs.py -q -x -k "frozen" 2>&1 | tail -5 | tee $O/u_tests.log
timeout 400 python tools/prefill_probe.py 2>&1 | grep -v Warn | tail -4 | tee $O/u_probe.log
