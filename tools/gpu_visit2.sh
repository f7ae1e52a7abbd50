#!/bin/bash
# visit: flash mode 6 correctness + perf, config tests, GELU big-tile A/B, bench A/B
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "flash" 2>&1 | tail -15 > $O/v2_flash_tests.log
echo "exit ${PIPESTATUS[0]}" >> $O/v2_flash_tests.log
if grep -q "exit 0" $O/v2_flash_tests.log; then
  timeout 300 python tools/gpu_check.py flashperf geluperf > $O/v2_perf.log 2>&1
  timeout 300 python bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --option flash_mode=6 > $O/v2_bench_f6.log 2>&1
  timeout 300 python bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --option flash_mode=6 --option gemm_big_gelu=1 > $O/v2_bench_f6_gelu.log 2>&1
fi
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -15 > $O/v2_config_tests.log
tail -8 $O/v2_flash_tests.log; cat $O/v2_perf.log; for f in $O/v2_bench_f6.log $O/v2_bench_f6_gelu.log; do tail -1 $f | cut -c1-400; done; tail -6 $O/v2_config_tests.log
