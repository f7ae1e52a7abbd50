#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "gemm" 2>&1 | tail -15 > $O/v6_gemm_tests.log
echo "exit ${PIPESTATUS[0]}" >> $O/v6_gemm_tests.log
timeout 300 python tools/gpu_check.py skinnyperf > $O/v6_skinnyperf.log 2>&1
if grep -q "exit 0" $O/v6_gemm_tests.log; then
  timeout 600 python tools/ab_bench.py base:tta_overlap=0 noskinny:tta_overlap=0,gemm_skinny=-1 base1::1 noskinny1:gemm_skinny=-1:1 > $O/v6_ab.log 2>&1
  timeout 600 python -m pytest tests/test_gpu_path.py -m gpu -q 2>&1 | tail -5 > $O/v6_path_tests.log
fi
tail -12 $O/v6_gemm_tests.log; grep -v amdgpu $O/v6_skinnyperf.log; cat $O/v6_ab.log; tail -3 $O/v6_path_tests.log
