#!/bin/bash
# A/B builds of the library on the same box: [ROUNDS=3] [BENCH_ARGS="..."] tools/ab.sh libA.so libB.so [libC.so ...]
# (HTM_LIB_OLDER_BUILD=1 lets _lib.py load a build of an older header: symbols it lacks are skipped)
N=${ROUNDS:-3}
for i in $(seq $N); do
  for L in "$@"; do
    v=$(HTM_LIB=$L HTM_LIB_OLDER_BUILD=1 timeout -k 10 120 python bench.py --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f steps/s  %.3f us/iteration' % (d['value'], d['config']['us_per_iteration']))")
    echo "$(basename $L) $v"
  done
done
