#!/bin/bash
# Run ON THE GPU BOX: the table of DESIGN.md §5 (other shapes on the same build).  tools/shape_table.sh > gpurun_out/shapes.txt
# (a bench step is a block of 8192 iterations: 6 timed steps per row are ~0.3 s of kernel time and more)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
row() {
  timeout -k 10 120 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
r=d['roofline']; b=d.get('roofline_batch64',{})
print('%-44s %10.0f steps/s %7.2f us/iter  k_mcmc %7.0f GB/s  k_full x64 %7.0f GB/s' % (d['config']['workload'][:44], d['value'], d['config']['us_per_iteration'], r['achieved'], b.get('achieved',0)))"
}
row --events 100 --stations 16 --chains 1 --steps 6 --warmup 2 --iters-per-step 8192
row --events 100 --stations 16 --chains 2 --steps 6 --warmup 2 --iters-per-step 8192
for c in 1 2 4 8 16 32; do row --chains $c --steps 6 --warmup 2 --iters-per-step 8192; done
row --events 10000 --stations 128 --chains 16 --steps 4 --warmup 1 --iters-per-step 2048
row --events 10000 --stations 128 --chains 16 --steps 4 --warmup 1 --iters-per-step 2048 --forward-precision fp32
