#!/bin/bash
# usage: tools/tile_sweep.sh [vth]: hot two-stream at fixed sort intervals, TILE order (VPIC_HIP_WINDOW=tile) against the wide row window
cd "$(dirname "$0")/.."; ulimit -c 0
vth=${1:-0.6}
for w in wide tile; do for si in 1 2 3 4 6 8 12; do
  echo -n "vth=$vth window=$w sort_interval=$si: "
  VPIC_HIP_WINDOW=$w python bench.py --config 1 --vth $vth --sort-interval $si --steps 24 --warmup 12 --no-cpu-baseline 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.2f G/s  avg_launch %.3f ms  frac %.3f  ms/step %.2f' % (d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step']))
    elif 'rror' in l: print(l.strip())"
done; done
