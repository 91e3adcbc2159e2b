set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r03_z_gpu_suite.log 2>&1 || { tail -40 gpurun_out/r03_z_gpu_suite.log; exit 1; }
tail -3 gpurun_out/r03_z_gpu_suite.log
