#!/bin/bash
# Round-end style visit: GPU pytest suite, smoke, bench line, rocprofv3 kernel stats + HBM counters of the same bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd $R
if [ -z "$SKIP_PYTEST" ]; then   # (SKIP_PYTEST=1: bench / profiles only -- the suite takes 12 of the visit's 14 minutes)
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -30 > $O/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> $O/pytest_gpu.log
fi
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" >> $O/smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.log 2>&1; echo "bench exit $?" >> $O/bench.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/prof $O/pmc_bench
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-train-step > $O/rocprof.log 2>&1
echo "rocprof exit $?" >> $O/rocprof.log
# volumes profiled per pass below: 2-stream run 1 warm-up + 5 repeats x 3 steps, one-stream run 1 + 2 x 3 = 23
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_bench/$c -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train-step > $O/pmc_$c.log 2>&1
  echo "pmc $c exit $?" >> $O/rocprof.log
done
cd $R
python tools/pmc_traffic.py $O/pmc_bench 23 > $O/traffic.json 2> $O/traffic.err
find $O/prof $O/pmc_bench -name "*kernel_trace*" -size +8M -delete 2>/dev/null
find $O/pmc_bench -name "*counter_collection*" -size +8M -delete 2>/dev/null
tail -4 $O/pytest_gpu.log; tail -2 $O/smoke.log; tail -2 $O/bench.log; tail -4 $O/rocprof.log; head -c 1500 $O/traffic.json
# (round 6: the parity record of the suite is gpurun_out/r06_parity.json)
# extras of round 3: decoder prefill probe, whole stage-1 step (also in the bench line), optimiser probe
timeout 300 python tools/prefill_probe.py 2>&1 | grep -v Warn | tail -3 > $O/prefill_probe.log
timeout 300 python tools/adamw_probe.py 2>&1 | grep -v Warn | tail -3 > $O/adamw_probe.log
cat $O/prefill_probe.log $O/adamw_probe.log
