# Run ON THE GPU BOX: several master workgroups (HTM_MB=1) against the single workgroup -- parity on rejection-heavy toy sizes
# (every rejection is an epoch change through memory) and timings.   bash tools/mb_check.sh <tag>
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T=${1:-r04_k}
run() { out=$1; shift; ( CHECK_MB=1 timeout -k 10 200 "$@" 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_$out.txt ) || echo "$out FAILED"; echo "== $out"; tail -n 11 gpurun_out/${T}_$out.txt; }
PIPE_CHECK_SZ=12 PIPE_CHECK_COOL=2 PIPE_CHECK_INT=3 run mb_rej16 python tools/pipe_check.py 16 64 64 3000 20000
PIPE_CHECK_SZ=20 PIPE_CHECK_COOL=2 PIPE_CHECK_INT=3 run mb_rej11 python tools/pipe_check.py 11 64 32 4000 20000
PIPE_CHECK_SZ=6 PIPE_CHECK_COOL=3 PIPE_CHECK_INT=7 run mb_rej13 python tools/pipe_check.py 13 300 128 3000 20000
run mb_16 python tools/pipe_check.py 16 1000 64 2000 50000
PIPE_CHECK_NOSLOG=1 run mb_c4_fp64 python tools/pipe_check.py 16 10000 128 300 3000
PIPE_CHECK_NOSLOG=1 PIPE_CHECK_PREC=fp32 run mb_c4_fp32 python tools/pipe_check.py 16 10000 128 300 3000
