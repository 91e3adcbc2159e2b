#!/bin/bash
# A/B two builds on the stand-alone batched full evaluation (64 models): tools/ab_full.sh libA.so libB.so [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    v=$(HTM_LIB=$L python bench.py --no-cpu-baseline --steps 2000 --warmup 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read())['roofline_batch64']; print('%.1f GB/s %.2f us' % (d['achieved'], d['avg_launch_us']))")
    echo "$(basename $L) $v"
  done
done
