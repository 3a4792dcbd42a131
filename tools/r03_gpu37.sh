#!/bin/bash
# does what runs before the 137 GB deck in the default bench line cost it anything?  default line against --no-second-config, interleaved
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
line() { python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1: value %.2f G/s ms/step %.2f (median %.2f) plain launch %.3f ms frac %.3f' % (d['value']/1e9, d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"; }
for rep in 1 2; do
python bench.py --no-cpu-baseline 2>&1 | line "default line     "
python bench.py --no-cpu-baseline --no-second-config 2>&1 | line "no second config "
done
