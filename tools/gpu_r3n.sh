#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_backward.py -q -x -k "adamw" 2>&1 | tail -5 > $O/n_tests.log
timeout 300 python tools/adamw_probe.py 2>&1 | grep -v Warn | tail -4 > $O/n_probe.log
timeout 600 python tools/train_step_full.py 3 2>&1 | tail -2 > $O/n_full.log
tail -3 $O/n_tests.log; cat $O/n_probe.log; tail -1 $O/n_full.log | cut -c1-400
