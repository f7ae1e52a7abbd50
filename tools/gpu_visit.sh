#!/bin/bash
# mid-round visit: selected GPU tests + bench line with the profiler-clock roofline (no CPU baseline / training legs)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/visit; rm -rf $O; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests -m gpu -q -k "${1:-flash or vit or config or path or gemm}" 2>&1 | tail -15 > $O/pytest.log
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-step ${BENCH_ARGS} > $O/bench.log 2>&1; echo "bench exit $?" >> $O/bench.log
cat $O/pytest.log; tail -3 $O/bench.log | cut -c1-3000
