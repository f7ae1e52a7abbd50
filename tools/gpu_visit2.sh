#!/bin/bash
# Visit 2: where does the ping-pong GEMM lose its time?  INM variants, ablations, PMC counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; O=$R/gpurun_out
: > $O/summary.log
run() { local n=$1 t=$2; shift 2; timeout $t "$@" > $O/$n.log 2>&1; echo "$n exit $?" >> $O/summary.log; }
run ppc_inm 300 python tools/gpu_check.py ppc:9,10,11,12
run ppperf2 500 python tools/gpu_check.py ppperf:few
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counters_all.txt 2>&1
grep -oE "\b(SQ|TA|TCP|TCC|GRBM|TD|SPI)_[A-Z0-9_]+" $O/counters_all.txt | sort -u > $O/counters.txt
wc -l $O/counters.txt >> $O/summary.log
pmc() { # tag what variant counters...
  local tag=$1 what=$2 v=$3; shift 3
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc/$tag -o p -- python $R/tools/prof_kernels.py $what 3 $v > $O/pmc_$tag.log 2>&1
  echo "pmc $tag exit $?" >> $O/summary.log
}
for cfg in "c:-1" "v7:7" "v3:3" "v9:9"; do
  t=${cfg%%:*}; v=${cfg##*:}
  for w in gemm4k gemm8k; do
    pmc ${w}_${t}_sq $w $v SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16
    pmc ${w}_${t}_lds $w $v SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_ADDR_CONFLICT
    pmc ${w}_${t}_ta $w $v TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
    pmc ${w}_${t}_mem $w $v FETCH_SIZE GRBM_GUI_ACTIVE
  done
done
cd $R
cat $O/summary.log
echo "=== ppc_inm"; grep -v "repeatable=True" $O/ppc_inm.log | tail -12; grep -c "repeatable=True" $O/ppc_inm.log
echo "=== ppperf2"; tail -12 $O/ppperf2.log
python - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/pmc"
for d in sorted(glob.glob(O + "/*")):
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in rows:
        k = r.get("Kernel_Name", "")[:40]
        if "gemm" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, c in agg.items():
        print(os.path.basename(d), k, {n: round(v / cnt[(k, n)]) for n, v in c.items()})
PY
