set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for wd in 0 1; do for pr in fp64 fp32; do
HTM_WIDE=$wd timeout -k 10 200 python bench.py --no-cpu-baseline --events 10000 --stations 128 --chains 16 --steps 4 --warmup 1 --iters-per-step 2048 --forward-precision $pr 2>gpurun_out/r03_j_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wide=$wd $pr', d['value'], d['config']['us_per_iteration'], d['roofline']['achieved'])"
done; done
HTM_WIDE=1 timeout -k 10 200 python bench.py --no-cpu-baseline --events 10000 --stations 64 --chains 8 --steps 4 --warmup 1 --iters-per-step 4096 2>gpurun_out/r03_j_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wide=1 10000x64x8', d['value'], d['config']['us_per_iteration'], d['roofline']['achieved'])"
HTM_WIDE=0 timeout -k 10 200 python bench.py --no-cpu-baseline --events 10000 --stations 64 --chains 8 --steps 4 --warmup 1 --iters-per-step 4096 2>gpurun_out/r03_j_err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wide=0 10000x64x8', d['value'], d['config']['us_per_iteration'], d['roofline']['achieved'])"
timeout -k 10 900 python -m pytest tests/test_gpu_chains.py tests/test_gpu_fp32.py -m gpu -x -q -k "c5 or configs4 or 10000 or fp32 or running_loglik" > gpurun_out/r03_j_tests.log 2>&1 || { tail -60 gpurun_out/r03_j_tests.log; exit 1; }
tail -3 gpurun_out/r03_j_tests.log
