#!/bin/bash
# rocprofv3 counter passes (counters + kernel trace only) for the hot kernels (-> profiles/rNN_kernel_pmc.json): flash
#   attention (double pipeline), big-tile GEMM on the ViT qkv shape and the SVR's packed q|k|v (256x192 tiles, two-stage and
#   deep forms; round 6: the drain forms the heuristic now picks for q|k|v and fc1 + GELU), the ring form on the SVR output projection, the 128^2 kernel on the fc1 shape, and
#   the 64^2 split-K kernel on the M = 256 query-side product of the TTA with cold weights.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/pmc2; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
run() {  # name, driver args..., then counter sets come from PASSES
  name=$1; shift
  i=0
  for ctrs in "${PASSES[@]}"; do
    i=$((i+1))
    timeout 60 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/${name}_p$i -o p -- python $R/tools/prof_kernels.py "$@" > $O/${name}_p$i.log 2>&1
    echo "$name pass $i exit $?"
  done
}
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS")
# Lessons of round 1 (25 GPU-minutes): "GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE" in ONE pass aborts rocprofv3 (signal 6) --
# collect them one per pass as tools/gpu_round.sh does; with SQ_INSTS_VALU added to the LDS set the profiled process
# never exited and every pass ran into its timeout (the counters were still written).  Keep the per-pass timeout short.
PASSES+=("FETCH_SIZE" "WRITE_SIZE")
run flash7pre flash 3 0 7 1
run qkv_bt192_deep gemm 3 24
run qkv_drain gemm 3 0
run svr_qkv_bt192_two_stage gemmsvr 8 21
run svr_qkv_bt192_deep gemmsvr 8 24
run svr_out_ring gemm4k 8 22
run fc1_gelu128 gemmmlp 3 -1
run fc1_gelu_bt256 gemmmlp 3 20
run fc1_gelu_drain gemmmlp 3 0
run tta_query_unsplit gemm256 16 0
run skinny64 gemm256old 16 0
run tokattn tokattn 5
run tokattn_4wave tokattn 5 1
# (unchanged kernels keep their round-3 rows in profiles/r03_kernel_pmc.json: flashbwd, prefillattn, kmajor_dw -- add them back here to refresh)
cd $R && python tools/pmc_kernels.py $O > $R/gpurun_out/kernel_pmc.json && cat $R/gpurun_out/kernel_pmc.json | head -80
