cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
HTM_LIB=hypotremormcmc_amd/lib/libhtm_hip_stamps.so timeout -k 10 200 python tools/flow_stamps.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_e_flow_stamps.txt
cat gpurun_out/r03_e_flow_stamps.txt
