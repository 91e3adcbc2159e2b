# Run ON THE GPU BOX: the lock-step paths after a change -- parity first, then the timings.   bash tools/r04_lock.sh <tag> [A/B library]
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T=${1:-r04_b}
ALT=$2
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests/test_gpu_chains.py -x -q -k "lockstep or torchworld or rccl or wraps or shared_prior" > gpurun_out/${T}_lock_tests.log 2>&1; rc=$?
tail -n 15 gpurun_out/${T}_lock_tests.log
[ $rc -eq 0 ] || exit $rc
fi
S=hypotremormcmc_amd/lib/libhtm_hip_stamps.so
if [ -f $S ]; then HTM_STAMPS_LOCK=1 timeout -k 10 200 python tools/flow_stamps.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_lock_stamps.txt; cat gpurun_out/${T}_lock_stamps.txt; fi
for L in hypotremormcmc_amd/lib/libhtm_hip.so $ALT; do
  HTM_LIB=$L python3 bench.py --force-lockstep --no-cpu-baseline > gpurun_out/${T}_bench_lockstep.json 2>gpurun_out/${T}_bench_lockstep.err
  python3 -c "import json; d=json.load(open('gpurun_out/${T}_bench_lockstep.json')); print('$L lock-step, 1 rank:', d['value'], d['config']['us_per_iteration'])"
  for n in 2 4; do
    ( HTM_LIB=$L HTM_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus $n --steps 4 --warmup 1 --iters-per-step 8192 --no-cpu-baseline > gpurun_out/${T}_rehearsal_$n.json 2>gpurun_out/${T}_rehearsal_$n.err || echo "rehearsal $n failed" )
    python3 -c "
import json
d=json.load(open('gpurun_out/${T}_rehearsal_$n.json')); print('$L rehearsal', d['n_gpus'], d['value'], d['config']['us_per_iteration'], d['config'].get('swap_transport','')[:40])"
  done
done
