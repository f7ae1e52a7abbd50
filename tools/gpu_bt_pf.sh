#!/bin/bash
# L2-prefetch variants of the four-wave big-tile K loops (builds under ab_pf/: distance in K tiles, operands) beside the repo's
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
(timeout 100 python tools/bt_epilogue_probe.py 2>&1 | grep -v amdgpu.ids
for d in ab_pf/*; do timeout 100 python tools/bt_epilogue_probe.py --root $d 2>&1 | grep -v amdgpu.ids; done
timeout 100 python tools/bt_epilogue_probe.py 2>&1 | grep -v amdgpu.ids) > $O/bt_pf_vit.log
(S="2048x4096x4096 1024x8192x4096 1792x8192x4096 2048x12288x4096"
timeout 100 python tools/bt_sweep.py --only "default,256x128 ring,256x192,256x256" $S 2>&1 | grep -v amdgpu.ids
for d in ab_pf/*; do timeout 100 python tools/bt_sweep.py --root $d --only "default,256x128 ring,256x192,256x256" $S 2>&1 | grep -v amdgpu.ids; done) > $O/bt_pf_tok.log
cat $O/bt_pf_vit.log $O/bt_pf_tok.log
