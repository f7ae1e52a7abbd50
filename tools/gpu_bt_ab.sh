#!/bin/bash
# Same-box A/B of two builds of the big-tile GEMM (ab_old/ = a copy of the package with another libu2tok_hip.so): the ViT's
# products and the tokenizer's, cold weights, each build measured twice in alternation.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
for i in 1 2; do
  timeout 120 python tools/bt_epilogue_probe.py --root ab_old 2>&1 | grep -v amdgpu.ids
  timeout 120 python tools/bt_epilogue_probe.py 2>&1 | grep -v amdgpu.ids
done > $O/bt_ab_vit.log
for i in 1 2; do
  timeout 150 python tools/bt_sweep.py --root ab_old --only "256x128 ring,256x192,256x256" 2048x4096x4096 1024x8192x4096 1792x8192x4096 2048x12288x4096 2>&1 | grep -v amdgpu.ids
  timeout 150 python tools/bt_sweep.py --only "default,256x128 ring,256x192,256x256" 2048x4096x4096 1024x8192x4096 1792x8192x4096 2048x12288x4096 1024x6144x4096 2>&1 | grep -v amdgpu.ids
done > $O/bt_ab_tok.log
cat $O/bt_ab_vit.log $O/bt_ab_tok.log
