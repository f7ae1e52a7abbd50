#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
mkdir -p $O; cd $R
timeout 600 python tools/determinism_probe.py > $O/v9_determinism.log 2>&1
grep -v amdgpu $O/v9_determinism.log
