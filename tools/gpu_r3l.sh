#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_ops.py -q -x 2>&1 | tail -15 > $O/l_tests.log
timeout 300 python tools/prefill_probe.py > $O/l_probe.log 2>&1
timeout 300 python tools/prefill_sweep.py "gemm_splitk=-1" "gemm_splitk=3" "gemm_splitk=4" "gemm_tile=64" > $O/l_sweep.log 2>&1
timeout 300 python tools/ab_bench.py base: base1::1 > $O/l_ab.log 2>&1
tail -5 $O/l_tests.log; grep -v Warning $O/l_probe.log | tail -4; tail -7 $O/l_sweep.log; tail -3 $O/l_ab.log
