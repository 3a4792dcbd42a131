#!/bin/bash
# passes per wavefront (VPIC_HIP_ITERS) against particles per cell:  tools/iters_sweep.sh "<bench args>" "<iters list>"
cd "$(dirname "$0")/.."
for it in $2; do
  echo -n "iters=$it  "
  VPIC_HIP_ITERS=$it python bench.py --no-cpu-baseline --steps 10 --warmup 3 $1 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('ms/step %.3f  pushes/s %.3e  roofline %.3f  launch %.3f ms' % (j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['avg_launch_ms']))"
done
