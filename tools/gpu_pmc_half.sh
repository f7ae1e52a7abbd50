R=$PWD; O=$R/gpurun_out/pmc_half; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in 24 28; do
  timeout 60 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/v$v -o p -- python $R/tools/prof_kernels.py gemmsvr 8 $v > $O/v$v.log 2>&1
  echo "variant $v exit $?"
done
cd $R
python - <<'PY'
import csv, glob, collections
for v in (24, 28):
    agg = collections.defaultdict(float); n = 0
    for f in glob.glob(f"gpurun_out/pmc_half/v{v}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_bt_kernel" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
    print(v, {k: round(val) for k, val in agg.items()})
PY
