#!/bin/bash
# One GPU-box visit: run each diagnostic section in its own process (a fault in one must not hide the others).
# Usage (through gpurun): bash tools/gpu_round.sh gemm ops attn modules perf
mkdir -p gpurun_out
: > gpurun_out/summary.log
for s in "$@"; do
  timeout 900 python tools/gpu_check.py "$s" > "gpurun_out/check_$s.log" 2>&1
  echo "section $s exit $?" >> gpurun_out/summary.log
done
cat gpurun_out/summary.log
for s in "$@"; do echo "=== $s"; tail -n 60 "gpurun_out/check_$s.log"; done
