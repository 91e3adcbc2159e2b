set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python3 bench.py --force-lockstep --no-cpu-baseline --steps 4 --warmup 1 --iters-per-step 8192 2>gpurun_out/r03_l_lock.err | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lock-step flow, 1 rank:', d['value'], d['config']['us_per_iteration'])" || { tail -5 gpurun_out/r03_l_lock.err; exit 1; }
HTM_FLOW_LOCK=0 timeout -k 10 300 python3 bench.py --force-lockstep --no-cpu-baseline --steps 4 --warmup 1 --iters-per-step 8192 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('lock-step barriers, 1 rank:', d['value'], d['config']['us_per_iteration'])"
timeout -k 10 900 python -m pytest tests/test_gpu_chains.py tests/test_gpu_fortran.py tests/test_bench_launcher.py -m gpu -x -q > gpurun_out/r03_l_tests.log 2>&1 || { tail -60 gpurun_out/r03_l_tests.log; exit 1; }
tail -3 gpurun_out/r03_l_tests.log
