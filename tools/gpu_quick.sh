#!/bin/bash
# quick visit: selected pytest -k filter + perf section
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "${1:-flash or vit or multiscale or transpose}" 2>&1 | tail -15
timeout 600 python tools/gpu_check.py ${2:-perf} 2>&1 | grep -v "gemm glds=0" | tail -40
