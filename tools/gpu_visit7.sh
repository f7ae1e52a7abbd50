#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_path.py -q -x -k "softmax or tokenizer or tta or full_path" 2>&1 | tail -4
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_ov.log 2>&1; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"row_ops[^}]*}' $O/bench_ov.log | head -8
