set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py tests/test_gpu_fp32.py -m gpu -x -q > gpurun_out/r03_k_fwd.log 2>&1 || { tail -40 gpurun_out/r03_k_fwd.log; exit 1; }
tail -2 gpurun_out/r03_k_fwd.log
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 --iters-per-step 8192 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline_batch64']['achieved'], d['roofline_batch64']['avg_launch_us'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 --iters-per-step 1024 --events 10000 --stations 128 --chains 16 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline_batch64']['achieved'], d['roofline_batch64']['avg_launch_us'])"
