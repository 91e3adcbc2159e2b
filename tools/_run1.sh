set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/log_gputests.txt 2>&1 || { tail -30 gpurun_out/log_gputests.txt; exit 1; }
tail -3 gpurun_out/log_gputests.txt
timeout -k 10 200 python -m pytest tests/test_gpu_forward.py -q -s -k "logarithm or square_root" 2>&1 | grep "htm_log" || true
ROUNDS=4 tools/ab.sh hypotremormcmc_amd/lib/libhtm_hip_fma.so hypotremormcmc_amd/lib/libhtm_hip.so > gpurun_out/log_ab.txt 2>&1
cat gpurun_out/log_ab.txt
ROUNDS=2 BENCH_ARGS="--events 10000 --stations 128 --chains 16 --steps 4 --warmup 1 --iters-per-step 2048" tools/ab.sh hypotremormcmc_amd/lib/libhtm_hip_fma.so hypotremormcmc_amd/lib/libhtm_hip.so > gpurun_out/log_ab_c5.txt 2>&1
cat gpurun_out/log_ab_c5.txt
ROUNDS=2 BENCH_ARGS="--force-lockstep" tools/ab.sh hypotremormcmc_amd/lib/libhtm_hip_fma.so hypotremormcmc_amd/lib/libhtm_hip.so > gpurun_out/log_ab_lock.txt 2>&1
cat gpurun_out/log_ab_lock.txt
