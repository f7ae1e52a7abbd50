#!/bin/bash
# rocprofv3 PMC passes (counters only + kernel trace, as the pool requires) for one kernel driver.
# usage: bash tools/gpu_pmc.sh <what> "<counters pass 1>" ["<counters pass 2>" ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; what=$1; shift
mkdir -p $R/gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/pmc/${what}_p$i -o p -- python $R/tools/prof_kernels.py $what 3 > $R/gpurun_out/pmc/${what}_p$i.log 2>&1
  echo "pass $i exit $?"
done
cd $R/gpurun_out/pmc; ls */ | head; for f in */p_counter_collection.csv; do echo "== $f"; head -1 $f; grep -E "flash_d64|gemm_bf16" $f | head -12; done
