#!/bin/bash
# bench line with every leg except the CPU baseline / training legs, plus the vendor yardstick
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/visit; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-train-step > $O/bench.log 2>&1; echo "bench exit $?" >> $O/bench.log
timeout 600 python tools/lib_yardstick.py 2>&1 | grep -v Warn > $O/yardstick.log
tail -3 $O/bench.log | cut -c1-200; cat $O/yardstick.log
