#!/bin/bash
# visit: backward tests, preprocess tests, interleaved A/B of launcher options
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_preprocess.py -m gpu -q 2>&1 | tail -40 > $O/v4_tests.log
echo "exit ${PIPESTATUS[0]}" >> $O/v4_tests.log
timeout 600 python tools/ab_bench.py base: gelu:gemm_big_gelu=1 noside:tta_overlap=0 s3::3 gelu_noside:gemm_big_gelu=1,tta_overlap=0 nobig:gemm_big=-1 s1::1 > $O/v4_ab.log 2>&1
tail -25 $O/v4_tests.log; cat $O/v4_ab.log
