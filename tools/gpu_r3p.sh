#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_prefill.py tests/test_gpu_configs.py -q -x 2>&1 | tail -6 > $O/p_tests.log
tail -3 $O/p_tests.log
timeout 300 python tools/bt_sweep.py 256x12288x4096 1024x4096x12288 1024x6144x4096 2>&1 | grep -v Warn | cut -c1-120 | tee $O/p_sweep.log
timeout 300 python tools/prefill_probe.py 2>&1 | grep -v Warn | tail -3 | tee $O/p_probe.log
timeout 300 python tools/ab_bench.py base: nosk:gemm_big_skinny=0 base1::1 nosk1:gemm_big_skinny=0:1 2>/dev/null | tee $O/p_ab.log
