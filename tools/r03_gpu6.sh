cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc1 $R/gpurun_out/pmc2 $R/gpurun_out/pmc3
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace -d $R/gpurun_out/pmc1 -o p --output-format csv -- python3 $R/tools/flow_pmc.py > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmc2 -o p --output-format csv -- python3 $R/tools/flow_pmc.py > $R/gpurun_out/pmc2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_INSTS_BRANCH --kernel-trace -d $R/gpurun_out/pmc3 -o p --output-format csv -- python3 $R/tools/flow_pmc.py > $R/gpurun_out/pmc3.log 2>&1
cd $R
tail -2 gpurun_out/pmc1.log
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc1","pmc2","pmc3"):
    agg=collections.defaultdict(float); cnt=collections.defaultdict(int)
    for fn in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            if "k_mcmc" in row["Kernel_Name"]:
                agg[row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[row["Counter_Name"]]+=1
    for k in sorted(agg): print(d, k, "%.0f total over %d launches -> per partial step (160080 steps): %.1f" % (agg[k], cnt[k], agg[k]/160080.0))
PY
