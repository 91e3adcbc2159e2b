set -e
cd $GRAFT_REPO_ROOT
ROUNDS=2 BENCH_ARGS="--steps 6 --warmup 2 --iters-per-step 16384" bash tools/ab.sh hypotremormcmc_amd/lib/libhtm_prev.so hypotremormcmc_amd/lib/libhtm_hip.so 2>&1
for L in hypotremormcmc_amd/lib/libhtm_prev.so hypotremormcmc_amd/lib/libhtm_hip.so; do
HTM_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --events 10000 --stations 128 --chains 16 --forward-precision fp32 --steps 4 --warmup 1 --iters-per-step 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 fp32', '$L'[-12:], '%.0f steps/s %.3f us' % (d['value'], d['config']['us_per_iteration']))"
HTM_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --chains 32 --steps 4 --warmup 1 --iters-per-step 8192 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('32 chains', '$L'[-12:], '%.0f steps/s %.3f us' % (d['value'], d['config']['us_per_iteration']))"
done
