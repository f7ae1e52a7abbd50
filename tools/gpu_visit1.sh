#!/bin/bash
# Visit 1 of this session: ping-pong GEMM variants (each in its own process, short timeouts: a barrier mismatch would
# hang), flash modes, then the GPU suite and a short bench.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
: > gpurun_out/summary.log
run() { # name timeout cmd...
  local n=$1 t=$2; shift 2
  timeout $t "$@" > gpurun_out/$n.log 2>&1
  echo "$n exit $?" >> gpurun_out/summary.log
}
run ppc4 240 python tools/gpu_check.py ppc:4
for v in 5 3 1 8 2 6 7; do run ppc$v 120 python tools/gpu_check.py ppc:$v; done
run ppperf 420 python tools/gpu_check.py ppperf
run flashperf 240 python tools/gpu_check.py flashperf
run attn 240 python tools/gpu_check.py attn
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > gpurun_out/pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/summary.log
run bench 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
cat gpurun_out/summary.log
for n in ppc4 ppc5 ppc3 ppc1 ppc8 ppc2 ppc6 ppc7; do echo "=== $n"; grep -v "repeatable=True" gpurun_out/$n.log | tail -8; grep -c "repeatable=True" gpurun_out/$n.log; done
echo "=== ppperf"; cat gpurun_out/ppperf.log | tail -25
echo "=== flashperf"; tail -16 gpurun_out/flashperf.log
echo "=== attn"; tail -14 gpurun_out/attn.log
echo "=== pytest"; cat gpurun_out/pytest_gpu.log
echo "=== bench"; tail -3 gpurun_out/bench.log
