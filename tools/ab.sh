#!/bin/bash
# A/B two builds of the library on the same box: tools/ab.sh libA.so libB.so [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    v=$(HTM_LIB=$L python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; print('%.0f' % json.loads(sys.stdin.read())['value'])")
    echo "$(basename $L) $v"
  done
done
