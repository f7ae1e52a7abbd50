#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
: > $O/summary.log
run() { local n=$1 t=$2; shift 2; timeout $t "$@" > $O/$n.log 2>&1; echo "$n exit $?" >> $O/summary.log; }
run ppc_split 300 python tools/gpu_check.py ppc:30,31,32
run ppperf3 500 python tools/gpu_check.py ppperf:few
cat $O/summary.log
echo "=== ppc_split"; grep -v "repeatable=True" $O/ppc_split.log | tail -12; grep -c "repeatable=True" $O/ppc_split.log
echo "=== ppperf3"; tail -10 $O/ppperf3.log
