#!/bin/bash
# A/B two builds at the C5 shape (10 000 x 128, 16 chains): tools/ab_c5.sh libA.so libB.so [rounds]
A=$1; B=$2; N=${3:-2}
for i in $(seq $N); do
  for L in $A $B; do
    v=$(HTM_LIB=$L python bench.py --no-cpu-baseline --events 10000 --stations 128 --chains 16 --steps 3000 --warmup 300 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f steps/s  k_mcmc %.0f GB/s' % (d['value'], d['roofline']['achieved']))")
    echo "$(basename $L) $v"
  done
done
