#!/bin/bash
# flash-attention visit: parity tests of the generated KV loops, per-launch timing and s_memtime phase table of modes 5 / 7,
# one SQ counter pass per mode (matrix-pipe busy, wait split).  Output: gpurun_out/flash_visit/*
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/flash_visit; rm -rf $O; mkdir -p $O; cd $R
timeout 600 python -m pytest tests -m gpu -q -k "flash or vit" 2>&1 | tail -15 > $O/pytest_flash.log
timeout 300 python tools/gpu_check.py flashperf 2>&1 | grep -v Warn > $O/flashperf.log
timeout 300 python tools/gpu_check.py flashtime 2>&1 | grep -v Warn > $O/flashtime.log
cd /tmp; export TMPDIR=/tmp
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
for m in ${MODES:-7_0 7_1}; do
  timeout 90 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc/flash${m}_p1 -o p -- python $R/tools/prof_kernels.py flash 3 0 ${m%_*} ${m#*_} > $O/pmc_flash$m.log 2>&1
  echo "pmc flash$m exit $?" >> $O/pmc.log
done
cd $R && python tools/pmc_kernels.py $O/pmc > $O/kernel_pmc.json 2>$O/pmc_kernels.err
find $O/pmc -name "*.csv" -size +4M -delete 2>/dev/null
cat $O/pytest_flash.log $O/flashperf.log $O/flashtime.log $O/pmc.log; cat $O/kernel_pmc.json
