# Run ON THE GPU BOX (through gpurun): what the driver runs at the end of a round -- the GPU test suite, smoke(), the bench under its
# command -- plus a two-rank rehearsal of the self-launching bench on the one GPU.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/final_bench.json')); print(d['metric'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_source'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'])"
HTM_BENCH_ONE_GPU=1 timeout -k 10 300 python3 bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2 ranks on one GPU (rehearsal):', d['n_gpus'], d['value'], d['config']['us_per_iteration'], d['config'].get('swap_transport'))"
