set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
ROUNDS=2 BENCH_ARGS="--steps 6 --warmup 2 --iters-per-step 16384" bash tools/ab.sh hypotremormcmc_amd/lib/libhtm_ts0.so hypotremormcmc_amd/lib/libhtm_hip.so 2>&1
ROUNDS=2 BENCH_ARGS="--steps 6 --warmup 2 --iters-per-step 16384" bash tools/ab.sh hypotremormcmc_amd/lib/libhtm_ts3.so hypotremormcmc_amd/lib/libhtm_hip.so 2>&1
