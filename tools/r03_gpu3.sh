set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_c_gputests.log 2>&1 || { tail -60 gpurun_out/r03_c_gputests.log; exit 1; }
tail -3 gpurun_out/r03_c_gputests.log
( timeout -k 10 500 python tools/stress_rejections.py 20000 > gpurun_out/r03_c_stress_rejections.txt 2>&1; echo "stress rc $?" )
tail -3 gpurun_out/r03_c_stress_rejections.txt
HTM_LIB=hypotremormcmc_amd/lib/libhtm_hip_stamps.so timeout -k 10 200 python tools/flow_stamps.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r03_c_flow_stamps.txt || true
cat gpurun_out/r03_c_flow_stamps.txt
for c in 1 2 4 8 16; do
timeout -k 10 200 python bench.py --no-cpu-baseline --chains $c --steps 6 --warmup 2 --iters-per-step 16384 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('chains $c', d['value'], d['config']['us_per_iteration'])"
done
