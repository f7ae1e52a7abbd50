#!/bin/bash
# K-loop ablations of the four-wave big-tile GEMM (measurement-only builds under ab_abl/ made by tools/mk_ab_build.sh ab_abl/NAME
# --ablate nodma,...; WRONG results): what is left of a
# product's time without the LDS-DMA / the fragment reads / the barrier / the MFMAs
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
(timeout 100 python tools/bt_epilogue_probe.py 2>&1 | grep -v amdgpu.ids
for d in ab_abl/*; do
  timeout 100 python tools/bt_epilogue_probe.py --root $d 2>&1 | grep -v amdgpu.ids
done
timeout 100 python tools/bt_epilogue_probe.py --big 20 2>&1 | grep -v amdgpu.ids
for d in ab_abl/*; do
  timeout 100 python tools/bt_epilogue_probe.py --root $d --big 20 2>&1 | grep -v amdgpu.ids
done) > $O/bt_ablations.log
cat $O/bt_ablations.log
