#!/bin/bash
# visit: full GPU suite, smoke, training-step probe, default bench
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd $R
rm -f $O/r02_parity.json
timeout 1500 python -m pytest tests -m gpu -q --durations=12 2>&1 | tail -40 > $O/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
timeout 600 python tools/train_step_probe.py > $O/train_step.log 2>&1; echo "exit $?" >> $O/train_step.log
timeout 900 python bench.py > $O/bench.log 2>&1; echo "bench exit $?" >> $O/bench.log
tail -30 $O/pytest_gpu.log; tail -2 $O/smoke.log; tail -3 $O/train_step.log; tail -2 $O/bench.log | cut -c1-600
