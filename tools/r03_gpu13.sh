cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r03_y
( timeout -k 10 300 python tools/time_fortran.py 1000000 > gpurun_out/${T}_fortran_timing.txt 2>&1 || true )
cat gpurun_out/${T}_fortran_timing.txt
rm -f gpurun_out/${T}_rehearsal_one_gpu.jsonl
for n in 2 4; do
  ( HTM_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus $n --steps 4 --warmup 1 --iters-per-step 8192 --no-cpu-baseline >> gpurun_out/${T}_rehearsal_one_gpu.jsonl 2>gpurun_out/${T}_rehearsal_$n.err || echo "rehearsal $n failed" )
done
python3 -c "
import json
for l in open('gpurun_out/${T}_rehearsal_one_gpu.jsonl'):
    d=json.loads(l); print('rehearsal', d['n_gpus'], d['value'], d['config']['us_per_iteration'], d['config'].get('swap_transport','')[:40], d['config'].get('us_per_iteration_by_rank'))"
HTM_LIB=hypotremormcmc_amd/lib/libhtm_hip_stamps.so timeout -k 10 200 python tools/flow_stamps.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_flow_stamps.txt || true
( timeout -k 10 400 python tools/soak_production.py > gpurun_out/${T}_soak_production.txt 2>&1; echo "soak rc $?" )
tail -5 gpurun_out/${T}_soak_production.txt
timeout -k 10 600 python -m pytest tests/test_gpu_chains.py tests/test_gpu_fortran.py -m gpu -x -q 2>&1 | tail -3
