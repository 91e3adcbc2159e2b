set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "
import sys; sys.path.insert(0,'.')
import __graft_entry__ as g; g.smoke()
" > gpurun_out/r03_f_smoke.log 2>&1 || { tail -30 gpurun_out/r03_f_smoke.log; exit 1; }
tail -1 gpurun_out/r03_f_smoke.log
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --iters-per-step 16384 2>gpurun_out/r03_f_bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('flow', d['value'], d['config']['us_per_iteration'])"
timeout -k 10 900 python -m pytest tests/test_gpu_chains.py -m gpu -x -q > gpurun_out/r03_f_chains.log 2>&1 || { tail -60 gpurun_out/r03_f_chains.log; exit 1; }
tail -3 gpurun_out/r03_f_chains.log
HTM_LIB=hypotremormcmc_amd/lib/libhtm_hip_stamps.so timeout -k 10 200 python tools/flow_stamps.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_f_flow_stamps.txt || true
cat gpurun_out/r03_f_flow_stamps.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --kernel-trace -d $R/gpurun_out/pmc1 -o p --output-format csv -- python3 $R/tools/flow_pmc.py > $R/gpurun_out/pmc1.log 2>&1
cd $R
grep "us/iteration" gpurun_out/pmc1.log
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc1",):
    agg=collections.defaultdict(float); cnt=collections.defaultdict(int)
    for fn in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(fn)):
            if "k_mcmc" in row["Kernel_Name"]:
                agg[row["Counter_Name"]]+=float(row["Counter_Value"]); cnt[row["Counter_Name"]]+=1
    for k in sorted(agg): print(d, k, "per partial step: %.1f" % (agg[k]/160080.0))
PY
