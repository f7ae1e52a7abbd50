#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_prefill.py tests/test_gpu_ops.py -q -x 2>&1 | tail -5 > $O/m_tests.log
for i in 1 2; do
  timeout 300 python gpurun_old/tools/ab_bench.py base: base1::1 2>/dev/null | sed 's/^/old /'
  timeout 300 python tools/ab_bench.py base: base1::1 2>/dev/null | sed 's/^/new /'
done | tee $O/m_ab.log
timeout 300 python tools/prefill_probe.py 2>&1 | grep -v Warn | tail -3 | tee $O/m_probe.log
tail -3 $O/m_tests.log
