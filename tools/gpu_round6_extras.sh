#!/bin/bash
# Round-6 extras beside tools/gpu_round.sh: counter rows of the hot kernels (incl. the drain forms), the vendor yardstick, the drain / skinny probes.
# Outputs under gpurun_out/r6x/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r6x; rm -rf $O; mkdir -p $O; cd $R
(timeout 300 python tools/lib_yardstick.py 2>&1 | grep -v Warn | tail -40) > $O/vendor_yardstick.log
(timeout 200 python tools/drain_probe.py 2>&1 | grep -v "Warn\|amdgpu.ids") > $O/drain_probe.log
(timeout 200 python tools/drain_probe.py --tail 2>&1 | grep -v "Warn\|amdgpu.ids") > $O/drain_probe_tail.log
(timeout 200 python tools/skinny_probe.py 2>&1 | grep -v "Warn\|amdgpu.ids") > $O/skinny_probe.log
tail -25 $O/vendor_yardstick.log; cat $O/drain_probe.log
bash tools/gpu_pmc2.sh > $O/pmc2.log 2>&1; tail -5 $O/pmc2.log | cut -c1-300
cp $R/gpurun_out/kernel_pmc.json $O/kernel_pmc.json 2>/dev/null
