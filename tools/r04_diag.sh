# Run ON THE GPU BOX (through gpurun): round-4 diagnostics of the three chain masters (free-running, lock-step, pipelined).
#   bash tools/r04_diag.sh <tag>
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T=${1:-r04_a}
S=hypotremormcmc_amd/lib/libhtm_hip_stamps.so
run() { out=$1; shift; ( timeout -k 10 240 "$@" 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_$out.txt ) || echo "$out failed"; echo "== $out"; tail -n 30 gpurun_out/${T}_$out.txt; }
run flow_stamps python tools/flow_stamps.py 8
HTM_STAMPS_LOCK=1 run lock_stamps python tools/flow_stamps.py 8
run pipe_stamps8 python tools/pipe_stamps.py 8
run pipe_stamps16 python tools/pipe_stamps.py 16
run pipe_trace8 python tools/pipe_trace.py 8
PIPE_CHECK_NOSLOG=1 run pipe_check8 python tools/pipe_check.py 8 1000 64 3000 200000
PIPE_CHECK_NOSLOG=1 run pipe_check16 python tools/pipe_check.py 16 1000 64 3000 100000
PIPE_CHECK_NOSLOG=1 run pipe_check32 python tools/pipe_check.py 32 1000 64 3000 50000
python3 bench.py --no-cpu-baseline > gpurun_out/${T}_bench.json 2>gpurun_out/${T}_bench.err; cat gpurun_out/${T}_bench.json
python3 bench.py --force-lockstep --no-cpu-baseline > gpurun_out/${T}_bench_lockstep.json 2>/dev/null; cat gpurun_out/${T}_bench_lockstep.json
