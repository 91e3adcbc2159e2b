cd $GRAFT_REPO_ROOT
A=hypotremormcmc_amd/lib/libhtm_hip_head.so; B=hypotremormcmc_amd/lib/libhtm_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_chains.py -x -q -k "rejection or single_rank or assorted or many_chains" 2>&1 | tail -2
ROUNDS=4 tools/ab.sh $A $B
ROUNDS=2 BENCH_ARGS="--force-lockstep" tools/ab.sh $A $B
ROUNDS=2 BENCH_ARGS="--chains 16 --steps 6 --warmup 2" tools/ab.sh $A $B
ROUNDS=2 BENCH_ARGS="--events 10000 --stations 128 --chains 16 --steps 4 --warmup 1 --iters-per-step 2048" tools/ab.sh $A $B
