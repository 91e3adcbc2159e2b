set -e
cd $GRAFT_REPO_ROOT
for v in vA vB; do
ROUNDS=2 BENCH_ARGS="--steps 6 --warmup 2 --iters-per-step 16384" bash tools/ab.sh hypotremormcmc_amd/lib/libhtm_prev.so hypotremormcmc_amd/lib/libhtm_$v.so 2>&1
for L in hypotremormcmc_amd/lib/libhtm_prev.so hypotremormcmc_amd/lib/libhtm_$v.so; do
HTM_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --force-lockstep --steps 4 --warmup 1 --iters-per-step 16384 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lockstep', '$L'[-12:], '%.0f steps/s %.3f us' % (d['value'], d['config']['us_per_iteration']))"
done; done
