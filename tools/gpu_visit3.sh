#!/bin/bash
# visit: backward tests, config tests, bench A/B on launcher options
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_backward.py -m gpu -q 2>&1 | tail -60 > $O/v3_backward_tests.log
echo "exit ${PIPESTATUS[0]}" >> $O/v3_backward_tests.log
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -q 2>&1 | tail -15 > $O/v3_config_tests.log
for opt in "gemm_big=-1" "gemm_big_gelu=1" "tta_overlap=0"; do
  timeout 300 python bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-roofline --option $opt 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$opt', j['value'], j['value_one_stream'])" >> $O/v3_ab.log 2>&1
done
timeout 300 python bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-roofline --streams 3 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('streams3', j['value'], j['value_one_stream'])" >> $O/v3_ab.log 2>&1
tail -40 $O/v3_backward_tests.log; tail -6 $O/v3_config_tests.log; cat $O/v3_ab.log
