#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "flash" 2>&1 | tail -8 > $O/v5_flash_tests.log
echo "exit ${PIPESTATUS[0]}" >> $O/v5_flash_tests.log
timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -q 2>&1 | tail -30 > $O/v5_backward_tests.log
timeout 300 python tools/gpu_check.py flashperf flashtime > $O/v5_flashperf.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline > $O/v5_bench.log 2>&1
timeout 600 python tools/prefill_probe.py > $O/v5_prefill.log 2>&1
tail -5 $O/v5_flash_tests.log; tail -12 $O/v5_backward_tests.log; cat $O/v5_flashperf.log | grep -v amdgpu; tail -1 $O/v5_bench.log | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['value_one_stream'], j['roofline']['frac'], j['roofline_attention']['avg_launch_us'], j['roofline_attention']['frac'])"; tail -2 $O/v5_prefill.log
