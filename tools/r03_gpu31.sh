set -e
cd $GRAFT_REPO_ROOT
for r in 1 2; do for L in hypotremormcmc_amd/lib/libhtm_prev.so hypotremormcmc_amd/lib/libhtm_hip.so; do
HTM_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --force-lockstep --steps 4 --warmup 1 --iters-per-step 16384 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lockstep', '$L'[-12:], '%.0f steps/s %.3f us' % (d['value'], d['config']['us_per_iteration']))"
done; done
for L in hypotremormcmc_amd/lib/libhtm_prev.so hypotremormcmc_amd/lib/libhtm_hip.so; do
HTM_LIB=$L HTM_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 4 --warmup 1 --iters-per-step 8192 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks one GPU', '$L'[-12:], '%.0f steps/s %.3f us' % (d['value'], d['config']['us_per_iteration']))"
done
