# Run ON THE GPU BOX (through gpurun): everything profiles/<tag>_* of a round comes from, in two calls (a call lasts at most 20 min):
#   bash tools/round_end.sh <tag> 1      rocprofv3 summaries + PMC passes of the bench, the bench line, lock-step / rehearsal lines, stamps, shapes
#   bash tools/round_end.sh <tag> 2      Fortran timings, production-length soak, stress runs
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=${1:-r04_z}
PART=${2:-1}
if [ "$PART" = "1" ]; then
bash tools/make_profiles.sh $T > gpurun_out/${T}_make_profiles.log 2>&1 || { tail -20 gpurun_out/${T}_make_profiles.log; exit 1; }
echo "profiles done"
python3 bench.py --force-lockstep --no-cpu-baseline > gpurun_out/${T}_bench_lockstep.json 2>/dev/null
python3 -c "import json; d=json.load(open('gpurun_out/${T}_bench_lockstep.json')); print('lock-step, 1 rank:', d['value'], d['config']['us_per_iteration'])"
rm -f gpurun_out/${T}_rehearsal_one_gpu.jsonl
for n in 2 4; do      # (a one-GPU box admits 6 processes on the GPU: no 8-rank rehearsal here)
  ( HTM_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus $n --steps 4 --warmup 1 --iters-per-step 8192 --no-cpu-baseline >> gpurun_out/${T}_rehearsal_one_gpu.jsonl 2>gpurun_out/${T}_rehearsal_$n.err || echo "rehearsal $n failed" )
done
python3 -c "
import json
for l in open('gpurun_out/${T}_rehearsal_one_gpu.jsonl'):
    d=json.loads(l); print('rehearsal', d['n_gpus'], d['value'], d['config']['us_per_iteration'], d['config'].get('swap_transport','')[:40])"
if [ -f hypotremormcmc_amd/lib/libhtm_hip_stamps.so ]; then
  HTM_LIB=hypotremormcmc_amd/lib/libhtm_hip_stamps.so timeout -k 10 200 python tools/flow_stamps.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_flow_stamps.txt || true
  HTM_STAMPS_LOCK=1 HTM_LIB=hypotremormcmc_amd/lib/libhtm_hip_stamps.so timeout -k 10 200 python tools/flow_stamps.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_lock_stamps.txt || true
fi
bash tools/shape_table.sh > gpurun_out/${T}_shapes.txt 2>&1
cat gpurun_out/${T}_shapes.txt
else
( timeout -k 10 300 python tools/time_fortran.py 1000000 > gpurun_out/${T}_fortran_timing.txt 2>&1 || true )
cat gpurun_out/${T}_fortran_timing.txt
( timeout -k 10 400 python tools/soak_production.py > gpurun_out/${T}_soak_production.txt 2>&1; echo "soak rc $?" )
tail -5 gpurun_out/${T}_soak_production.txt
( timeout -k 10 420 python tools/stress_rejections.py 20000 > gpurun_out/${T}_stress_rejections.txt 2>&1; echo "stress rc $?" )
tail -1 gpurun_out/${T}_stress_rejections.txt
fi
