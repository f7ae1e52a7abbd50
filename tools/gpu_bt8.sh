#!/bin/bash
# eight-wave big-tile forms (variants 30 / 31 / 32) beside the four-wave ones: the ViT's products and the tokenizer's, cold weights
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
(for v in 0 21 31 20 30 32 22; do
  a=""; [ $v != 0 ] && a="--big $v"
  timeout 100 python tools/bt_epilogue_probe.py $a 2>&1 | grep -v amdgpu.ids
done) > $O/bt8_vit.log
timeout 250 python tools/bt_sweep.py --only "default,256x128 ring,w8 256x256,w8 256x192,w8 256x128,256x192,256x256" 2048x4096x4096 1024x8192x4096 1792x8192x4096 2048x12288x4096 1024x6144x4096 4096x4096x4096 8192x8192x8192 2>&1 | grep -v amdgpu.ids > $O/bt8_tok.log
cat $O/bt8_vit.log $O/bt8_tok.log
