set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in 8 16 32; do for L in hypotremormcmc_amd/lib/libhtm_prev.so hypotremormcmc_amd/lib/libhtm_hip.so; do
HTM_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --chains $c --steps 4 --warmup 1 --iters-per-step 8192 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$c chains', '$L'[-12:], '%.0f steps/s %.3f us' % (d['value'], d['config']['us_per_iteration']))"
done; done
timeout -k 10 120 python bench.py --no-cpu-baseline --force-lockstep --steps 4 --warmup 1 --iters-per-step 8192 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lockstep', '%.0f steps/s %.3f us' % (d['value'], d['config']['us_per_iteration']))"
timeout -k 10 120 python bench.py --no-cpu-baseline --events 10000 --stations 128 --chains 16 --steps 4 --warmup 1 --iters-per-step 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 fp64', '%.0f steps/s %.3f us' % (d['value'], d['config']['us_per_iteration']))"
timeout -k 10 900 python -m pytest tests/test_gpu_chains.py -m gpu -x -q > gpurun_out/r03_n_chains.log 2>&1 || { tail -40 gpurun_out/r03_n_chains.log; exit 1; }
tail -2 gpurun_out/r03_n_chains.log
