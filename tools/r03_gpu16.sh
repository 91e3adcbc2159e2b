set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_chains.py tests/test_gpu_fortran.py tests/test_bench_launcher.py tests/test_gpu_fp32.py -m gpu -x -q -k "torchworld or mpi or bench_two or configs4" > gpurun_out/r03_m_tests.log 2>&1 || { tail -60 gpurun_out/r03_m_tests.log; exit 1; }
tail -3 gpurun_out/r03_m_tests.log
HTM_LIB=hypotremormcmc_amd/lib/libhtm_hip_stamps.so timeout -k 10 200 python tools/flow_stamps.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_z_flow_stamps.txt || true
cat gpurun_out/r03_z_flow_stamps.txt
