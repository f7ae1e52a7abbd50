#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
for ns in 1 2 3; do
timeout 600 python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-roofline --streams $ns > $O/bench_s$ns.log 2>&1; echo "streams $ns:" $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_s$ns.log | head -2)
done
timeout 600 python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-roofline --streams 2 --batch 2 > $O/bench_b2.log 2>&1; echo "streams 2 batch 2:" $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_b2.log | head -2)
timeout 600 python bench.py --steps 30 --warmup 4 --no-cpu-baseline --no-roofline --streams 1 --batch 2 > $O/bench_b2s1.log 2>&1; echo "streams 1 batch 2:" $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench_b2s1.log | head -2)
tail -2 $O/bench_b2.log | cut -c1-300
