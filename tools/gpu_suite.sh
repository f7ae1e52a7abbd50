#!/bin/bash
# Round-end style visit: GPU pytest suite, smoke, bench line, rocprofv3 kernel stats of the same bench command.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/pytest_gpu.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/bench.log 2>&1; echo "bench exit $?" >> gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof exit $?" >> $R/gpurun_out/rocprof.log
cd $R
find gpurun_out/prof -name "*kernel_trace*" -size +20M -delete 2>/dev/null
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -3; tail -3 gpurun_out/bench.log; tail -3 gpurun_out/rocprof.log; ls -la gpurun_out/prof/* | head
