set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
A=hypotremormcmc_amd/lib/${BASE:-libhtm_hip_karg.so}
B=hypotremormcmc_amd/lib/libhtm_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_chains.py -x -q > gpurun_out/log_gputests.txt 2>&1 || { tail -30 gpurun_out/log_gputests.txt; exit 1; }
tail -2 gpurun_out/log_gputests.txt
ROUNDS=4 tools/ab.sh $A $B 2>&1 | tee gpurun_out/log_ab.txt
ROUNDS=2 BENCH_ARGS="--force-lockstep" tools/ab.sh $A $B 2>&1 | tee gpurun_out/log_ab_lock.txt
ROUNDS=2 BENCH_ARGS="--chains 1 --steps 6 --warmup 2" tools/ab.sh $A $B 2>&1 | tee gpurun_out/log_ab_c1.txt
