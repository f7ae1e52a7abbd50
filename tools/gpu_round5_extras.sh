#!/bin/bash
# Round-5 extras beside tools/gpu_round.sh: the tokenizer-attention probes (both forms, timeline), the counter rows of the hot
# kernels, the vendor yardstick.  Outputs under gpurun_out/r5x/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r5x; rm -rf $O; mkdir -p $O; cd $R
(timeout 200 python tools/tokattn_probe.py 1 1 2>&1 | grep -v Warn | tail -9) > $O/tokattn_probe_8wave.log
(timeout 200 python tools/tokattn_probe.py 1 0 2>&1 | grep -v Warn | tail -9) > $O/tokattn_probe_4wave.log
(timeout 100 python tools/tokattn_probe.py timed 1 2>&1 | grep -v Warn | tail -9) > $O/tokattn_timeline.log
(timeout 300 python tools/lib_yardstick.py 2>&1 | grep -v Warn | tail -40) > $O/vendor_yardstick.log
cat $O/tokattn_probe_8wave.log $O/tokattn_timeline.log | cut -c1-400; tail -25 $O/vendor_yardstick.log
bash tools/gpu_pmc2.sh > $O/pmc2.log 2>&1; tail -5 $O/pmc2.log | cut -c1-300
cp $R/gpurun_out/kernel_pmc.json $O/kernel_pmc.json 2>/dev/null
