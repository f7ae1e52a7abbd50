#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
timeout 300 python tools/gpu_check.py pptime > $O/pptime.log 2>&1; echo "pptime exit $?"
tail -14 $O/pptime.log
