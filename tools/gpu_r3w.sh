#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm" 2>&1 | tail -3 | tee $O/w_tests.log
for i in 1 2; do
  timeout 120 python tools/bt_epilogue_probe.py --root gpurun_old 2>&1 | grep TF
  timeout 120 python tools/bt_epilogue_probe.py 2>&1 | grep TF
done | tee $O/w_probe.log
timeout 300 python gpurun_old/tools/ab_bench.py base: base1::1 2>/dev/null | sed 's/^/old /' | tee $O/w_ab.log
timeout 300 python tools/ab_bench.py base: base1::1 2>/dev/null | sed 's/^/new /' | tee -a $O/w_ab.log
