set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_chains.py tests/test_gpu_fortran.py tests/test_bench_launcher.py tests/test_gpu_fp32.py -m gpu -x -q > gpurun_out/r03_l_tests.log 2>&1 || { tail -60 gpurun_out/r03_l_tests.log; exit 1; }
tail -3 gpurun_out/r03_l_tests.log
rm -f gpurun_out/r03_l_rehearsal.jsonl
for n in 2 4; do
  ( HTM_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus $n --steps 4 --warmup 1 --iters-per-step 8192 --no-cpu-baseline >> gpurun_out/r03_l_rehearsal.jsonl 2>gpurun_out/r03_l_rehearsal_$n.err || echo "rehearsal $n failed" )
done
python3 -c "
import json
for l in open('gpurun_out/r03_l_rehearsal.jsonl'):
    d=json.loads(l); print('rehearsal', d['n_gpus'], d['value'], d['config']['us_per_iteration'], d['config'].get('swap_transport','')[:40])"
