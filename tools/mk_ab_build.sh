#!/bin/bash
# Builds a SECOND copy of the package with a differently generated big-tile K loop, for same-box A/B runs on the GPU
# (tools/gpu_bt_ab.sh, tools/gpu_bt_abl.sh import it with --root):
#     tools/mk_ab_build.sh DEST_DIR [generator flags ...]      e.g.  tools/mk_ab_build.sh ab_abl/nodma --ablate nodma
# DEST_DIR/u2tokenizer_amd gets the Python files and its own lib/libu2tok_hip.so (git-ignored; it travels with gpurun).
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=$1; shift
T=$(mktemp -d); cp -r $R/u2tokenizer_amd $T/; cp -r $R/include $T/
(cd $T/u2tokenizer_amd/csrc && python $R/tools/gen_gemm_bt_asm.py "$@" > gemm_bt_asm.inc && rm -f build/gemm_bt.o && make -j3 ../lib/libu2tok_hip.so > /dev/null 2>&1)
rm -rf $R/$D; mkdir -p $R/$D; cp -r $T/u2tokenizer_amd $R/$D/; rm -rf $R/$D/u2tokenizer_amd/csrc $R/$D/u2tokenizer_amd/__pycache__ $T
echo "$D: $(ls -la $R/$D/u2tokenizer_amd/lib/libu2tok_hip.so | awk '{print $5}') bytes"
