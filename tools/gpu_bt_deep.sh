#!/bin/bash
# deep forms of the big-tile GEMM (variants 23-26) beside the two-stage ones: the ViT's products and the tokenizer's, cold operands
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
(for v in 0 23 24 20 25 26 0 23; do
  a=""; [ $v != 0 ] && a="--big $v"
  timeout 100 python tools/bt_epilogue_probe.py $a 2>&1 | grep -v amdgpu.ids
done) > $O/bt_deep_vit.log
timeout 250 python tools/bt_sweep.py --only "default,256x128 ring,192 deepA,192 deepB,256 deepA,256 deepB,256x192,256x256" 2048x4096x4096 1024x8192x4096 1792x8192x4096 2048x12288x4096 4096x4096x4096 8192x8192x8192 2>&1 | grep -v amdgpu.ids > $O/bt_deep_tok.log
cat $O/bt_deep_vit.log $O/bt_deep_tok.log
