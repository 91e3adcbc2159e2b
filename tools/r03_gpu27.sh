set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
HTM_WIDE=1 timeout -k 10 120 python bench.py --no-cpu-baseline --events 10000 --stations 128 --chains 16 --forward-precision fp32 --steps 4 --warmup 1 --iters-per-step 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 fp32 wide', '%.0f steps/s %.3f us' % (d['value'], d['config']['us_per_iteration']))"
timeout -k 10 1000 python -m pytest tests/test_gpu_fp32.py tests/test_gpu_forward.py tests/test_gpu_chains.py -m gpu -x -q > gpurun_out/r03_s_tests.log 2>&1 || { tail -40 gpurun_out/r03_s_tests.log; exit 1; }
tail -2 gpurun_out/r03_s_tests.log
