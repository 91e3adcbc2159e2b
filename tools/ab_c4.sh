# Run ON THE GPU BOX: same-box A/B of two library builds at the configs[4] per-GPU shape (fp64 and fp32) and at configs[2].
#   bash tools/ab_c4.sh libA.so libB.so
cd ${GRAFT_REPO_ROOT:-.}
A=${1:-hypotremormcmc_amd/lib/libhtm_hip.so}; B=${2:-hypotremormcmc_amd/lib/libhtm_prev.so}
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-18s %-14s %9.0f steps/s %7.3f us' % ('$1', '$2'.split('/')[-1], d['value'], d['config']['us_per_iteration']))"; }
for i in 1 2; do
for L in $A $B; do
  HTM_LIB=$L HTM_LIB_OLDER_BUILD=1 timeout -k 10 120 python bench.py --no-cpu-baseline --events 10000 --stations 128 --chains 16 --steps 4 --warmup 1 --iters-per-step 1024 2>/dev/null | line "10000x128x16 fp64" $L
  HTM_LIB=$L HTM_LIB_OLDER_BUILD=1 timeout -k 10 120 python bench.py --no-cpu-baseline --events 10000 --stations 128 --chains 16 --forward-precision fp32 --steps 4 --warmup 1 --iters-per-step 1024 2>/dev/null | line "10000x128x16 fp32" $L
  HTM_LIB=$L HTM_LIB_OLDER_BUILD=1 timeout -k 10 120 python bench.py --no-cpu-baseline --steps 6 --warmup 2 --iters-per-step 16384 2>/dev/null | line "1000x64x8" $L
done
done
