set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "
import sys; sys.path.insert(0,'.')
import __graft_entry__ as g; g.smoke()
" > gpurun_out/r03_h_smoke.log 2>&1 || { tail -30 gpurun_out/r03_h_smoke.log; exit 1; }
tail -1 gpurun_out/r03_h_smoke.log
ROUNDS=2 BENCH_ARGS="--steps 6 --warmup 2 --iters-per-step 16384" bash tools/ab.sh hypotremormcmc_amd/lib/libhtm_baselog.so hypotremormcmc_amd/lib/libhtm_hip.so 2>&1 | tee gpurun_out/r03_h_ab.txt
timeout -k 10 900 python -m pytest tests/test_gpu_chains.py -m gpu -x -q > gpurun_out/r03_h_chains.log 2>&1 || { tail -60 gpurun_out/r03_h_chains.log; exit 1; }
tail -3 gpurun_out/r03_h_chains.log
HTM_LIB=hypotremormcmc_amd/lib/libhtm_hip_stamps.so timeout -k 10 200 python tools/flow_stamps.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_h_flow_stamps.txt || true
cat gpurun_out/r03_h_flow_stamps.txt
