#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for ab in "" 1 ""; do
  if [ -n "$ab" ]; then export U2_BT_ABLATE=1; else unset U2_BT_ABLATE; fi
  timeout 120 python tools/bt_epilogue_probe.py 2>&1 | grep TF | sed "s/^/ablate=${ab:-0} /"
done | tee $O/v_ablate.log
