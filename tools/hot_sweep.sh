#!/bin/bash
# push / step rates of the two-stream deck as the plasma gets hotter, for several sort intervals
#   tools/hot_sweep.sh "<vth list>" "<sort interval list>"   (negative interval: adaptive with that upper bound)
cd "$(dirname "$0")/.."
VTHS=${1:-"0.02 0.1 0.24 0.6"}; SIS=${2:-"10 5 2 1"}
for vth in $VTHS; do
  for si in $SIS; do
    echo -n "vth=$vth sort_interval=$si  "
    python bench.py --no-cpu-baseline --steps ${STEPS:-20} --warmup ${WARMUP:-5} --vth $vth --sort-interval $si 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('ms/step %.3f  pushes/s %.3e  roofline %.3f' % (j['ms_per_step'], j['value'], j['roofline']['frac']))"
  done
done
