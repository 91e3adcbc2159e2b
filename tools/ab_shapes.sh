# Run ON THE GPU BOX (through gpurun): same-box A/B of two library builds over the shapes DESIGN.md section 5 quotes.
#   bash tools/ab_shapes.sh hypotremormcmc_amd/lib/libhtm_prev.so hypotremormcmc_amd/lib/libhtm_hip.so [ranks]
# configs[2] (two rounds, alternating), 32 chains, one lock-step rank, the configs[4] per-GPU shape in fp64 and fp32; with a
# third argument also `ranks` lock-step ranks sharing the GPU (HTM_BENCH_ONE_GPU=1).  Every line: shape, library, steps/s, us per iteration.
set -e
cd ${GRAFT_REPO_ROOT:-.}
A=$1; B=$2; R=${3:-0}
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-18s %-14s %9.0f steps/s %7.3f us' % ('$1', '$2'.split('/')[-1], d['value'], d['config']['us_per_iteration']))"; }
ROUNDS=2 BENCH_ARGS="--steps 6 --warmup 2 --iters-per-step 16384" bash tools/ab.sh $A $B 2>&1
for L in $A $B; do
  HTM_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --chains 32 --steps 4 --warmup 1 --iters-per-step 8192 2>/dev/null | line "32 chains" $L
  HTM_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --force-lockstep --steps 4 --warmup 1 --iters-per-step 16384 2>/dev/null | line "lock-step 1 rank" $L
  HTM_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --events 10000 --stations 128 --chains 16 --steps 4 --warmup 1 --iters-per-step 1024 2>/dev/null | line "10000x128x16 fp64" $L
  HTM_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --events 10000 --stations 128 --chains 16 --forward-precision fp32 --steps 4 --warmup 1 --iters-per-step 1024 2>/dev/null | line "10000x128x16 fp32" $L
  if [ "$R" -gt 1 ]; then
    HTM_LIB=$L HTM_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus $R --steps 4 --warmup 1 --iters-per-step 8192 --no-cpu-baseline 2>/dev/null | line "$R ranks, one GPU" $L
  fi
done
