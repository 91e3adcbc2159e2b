cd $GRAFT_REPO_ROOT
for b in 512 1024 2048 4096; do
HTM_FULL_BLOCKS=$b timeout -k 10 300 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --iters-per-step 4096 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('blocks $b', d['roofline_batch64']['achieved'], d['roofline_batch64']['avg_launch_us'])"
done
