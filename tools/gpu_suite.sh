# Run ON THE GPU BOX (through gpurun): the whole `-m gpu` suite in one process, log under gpurun_out/.
set -e
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
T=${1:-suite}
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gpu_suite.log 2>&1 || { tail -40 gpurun_out/${T}_gpu_suite.log; exit 1; }
tail -3 gpurun_out/${T}_gpu_suite.log
