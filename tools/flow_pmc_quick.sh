# Run ON THE GPU BOX: one counter pass of tools/flow_pmc.py (vector / scalar / LDS / scalar-memory instructions per step) -- the quick
# check of an instruction-count change.  bash tools/flow_pmc_quick.sh
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out; mkdir -p $OUT; rm -rf $OUT/pmc_quick
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc_quick -o p --output-format csv -- python3 $ROOT/tools/flow_pmc.py 20000 > $OUT/pmc_quick.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]; steps = 20010 * 8; tot = {}
for f in glob.glob(os.path.join(out, "pmc_quick", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_mcmc" in r["Kernel_Name"]: tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print("  ".join("%s %.1f" % (c, tot[c] / steps) for c in sorted(tot)))
print([l for l in open(os.path.join(out, "pmc_quick.log")).read().splitlines() if l.startswith("us/iteration")])
PY
