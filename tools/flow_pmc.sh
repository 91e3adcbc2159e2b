# Run ON THE GPU BOX (through gpurun): instruction counters of the free-running master alone (tools/flow_pmc.py), three
# counter passes (counters in their own runs, --kernel-trace only), summary to gpurun_out/<tag>_flow_pmc.txt.
#   bash tools/flow_pmc.sh r03_z
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
T=${1:-flow}
mkdir -p $OUT
N=20000
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace -d $OUT/pmc_${T}_1 -o p --output-format csv -- python3 $ROOT/tools/flow_pmc.py $N > $OUT/pmc_${T}_1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVE_CYCLES --kernel-trace -d $OUT/pmc_${T}_2 -o p --output-format csv -- python3 $ROOT/tools/flow_pmc.py $N > $OUT/pmc_${T}_2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d $OUT/pmc_${T}_3 -o p --output-format csv -- python3 $ROOT/tools/flow_pmc.py $N > $OUT/pmc_${T}_3.log 2>&1
python3 - $OUT $T $N <<'PY'
import csv, glob, os, sys
out, tag, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
steps = (n + 10) * 8
lines = ["# Instruction counters of the free-running master ALONE (tools/flow_pmc.sh: tools/flow_pmc.py under rocprofv3 --pmc, three passes,",
         "# --kernel-trace): 1 000 events x 64 stations, 8 chains, hypocentre proposals only (no full evaluations after iteration 1), ONE worker",
         "# block (one polling wave); %d iterations = %d partial-update steps in the counted k_mcmc launches." % (n + 10, steps)]
for k in (1, 2, 3):
    tot = {}
    for f in glob.glob(os.path.join(out, "pmc_%s_%d" % (tag, k), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_mcmc" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for c in sorted(tot):
        lines.append("pass %d  %-22s %16.0f  -> per partial step: %9.1f" % (k, c, tot[c], tot[c] / steps))
    log = open(os.path.join(out, "pmc_%s_%d.log" % (tag, k))).read()
    for l in log.splitlines():
        if l.startswith("us/iteration"): lines.append("pass %d  %s (under the profiler)" % (k, l))
open(os.path.join(out, "%s_flow_pmc.txt" % tag), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
