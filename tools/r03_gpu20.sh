set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q > gpurun_out/r03_p_tests.log 2>&1 || { tail -40 gpurun_out/r03_p_tests.log; exit 1; }
tail -2 gpurun_out/r03_p_tests.log
for L in hypotremormcmc_amd/lib/libhtm_prev.so hypotremormcmc_amd/lib/libhtm_hip.so; do for r in 1 2; do
HTM_LIB=$L timeout -k 10 200 python bench.py --no-cpu-baseline --events 1000 --stations 64 --steps 2 --warmup 1 --iters-per-step 8192 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); b=d['roofline_batch64']; print('1000x64', '$L'[-12:], 'k_full x64 %.2f us frac %.3f' % (b['avg_launch_us'], b['frac']))"
done; done
