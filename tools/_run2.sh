set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
rocprofv3 -L > /tmp/avail.txt 2>&1 || true
(grep -o "SQ_[A-Z_0-9]*" /tmp/avail.txt || true) | sort -u | tr '\n' ' ' > gpurun_out/pmc/sq_names.txt
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  HTM_PERSIST=0 timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc/p$i -o p --output-format csv -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 1 --iters-per-step 512 > gpurun_out/pmc/p$i.log 2>&1 || { tail -5 gpurun_out/pmc/p$i.log; }
  tail -2 gpurun_out/pmc/p$i.log
done
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('/tmp/pmc/p*/')):
    for fn in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
        for r in csv.DictReader(open(fn)):
            k=r['Kernel_Name'][:40]; acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
        for k,v in acc.items():
            if 'k_step' in k or 'k_full' in k: print(d, k, dict(v))
PY
du -sh gpurun_out /tmp/pmc
