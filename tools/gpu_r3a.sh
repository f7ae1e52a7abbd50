#!/bin/bash
# round 3, visit A: new parity tests + fused tokenizer attention (correctness, probe, A/B in the pipeline)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "tok_attention" 2>&1 | tail -15 > $O/a_ops.log
timeout 600 python tools/tokattn_probe.py > $O/a_probe.log 2>&1
timeout 900 python -m pytest tests/test_gpu_path.py tests/test_checkpoint.py -m gpu -q 2>&1 | tail -15 > $O/a_path.log
timeout 900 python -m pytest tests/test_gpu_configs.py -q -k "teacher_forced or one_layer or config3" 2>&1 | tail -25 > $O/a_cfg.log
timeout 900 python -m pytest tests/test_gpu_backward.py -q -k "config4 or zero1 or linear_fn or tokenizer_gradients_vs_oracle" 2>&1 | tail -25 > $O/a_bwd.log
timeout 600 python tools/ab_bench.py base: noflash:tok_flash=0 base1::1 noflash1:tok_flash=0:1 > $O/a_ab.log 2>&1
for f in a_ops a_probe a_path a_cfg a_bwd a_ab; do echo "== $f"; grep -v amdgpu.ids $O/$f.log | tail -30; done
