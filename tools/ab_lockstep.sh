#!/bin/bash
# A/B two builds on the lock-step path (one rank, real RCCL): tools/ab_lockstep.sh libA.so libB.so [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    v=$(HTM_LIB=$L python bench.py --force-lockstep --no-cpu-baseline --steps 8000 --warmup 1000 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f %.2f us/iter' % (d['value'], 1e3*d['ms_per_step']))")
    echo "$(basename $L) $v"
  done
done
