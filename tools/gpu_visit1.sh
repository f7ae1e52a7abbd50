#!/bin/bash
# visit: full GPU test suite with durations, bench line, rocprofv3 kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd $R
rm -f $O/r02_parity.json
timeout 1500 python -m pytest tests -m gpu -q --durations=20 2>&1 | tail -60 > $O/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.log 2>&1; echo "bench exit $?" >> $O/bench.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline > $O/rocprof.log 2>&1
echo "rocprof exit $?" >> $O/rocprof.log
find $O/prof -name "*kernel_trace*" -size +8M -delete 2>/dev/null
cd $R
tail -45 $O/pytest_gpu.log; tail -3 $O/bench.log | cut -c1-1500; tail -2 $O/rocprof.log
