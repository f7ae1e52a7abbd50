#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "tok_attention" 2>&1 | tail -5 > $O/c_ops.log
timeout 300 python tools/tokattn_probe.py 1 > $O/c_probe1.log 2>&1
timeout 300 python tools/tokattn_probe.py 2 > $O/c_probe2.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_c
for v in 1 2; do
  i=0
  for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/pmc_c/v${v}_p$i -o p -- python $R/tools/tokattn_probe.py pmc $v > $O/c_pmc_v${v}_p$i.log 2>&1
    echo "pmc v$v pass $i exit $?"
  done
done
cd $R
for f in c_ops c_probe1 c_probe2; do echo "== $f"; grep -v amdgpu.ids $O/$f.log | tail -12; done
python - <<'PY'
import csv, glob, collections
for v in (1, 2):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"gpurun_out/pmc_c/v{v}_p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "tok_attn_kernel" in r.get("Kernel_Name", ""):
                a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f"== layout {v}:", {k: round(x[0] / max(x[1], 1)) for k, x in sorted(agg.items())})
PY
find $O/pmc_c -name "*.csv" -size +2M -delete 2>/dev/null
