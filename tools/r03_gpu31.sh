#!/bin/bash
# the sort inside the push (the default) against sort + push (VPIC_HIP_SORT_IN_PUSH=0), interleaved on one box
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04d; mkdir -p $O

line() { python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); s=d.get('advance_p_sorting') or {}
        print('$1: value %.2f G/s ms/step %.2f plain launch %.3f ms frac %.3f  sorting launch %s ms' % (d['value']/1e9, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], s.get('avg_launch_ms')))"; }
for cfg in "" "--config 1"; do
echo "== ${cfg:-config 2}"
for rep in 1 2 3; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-config $cfg 2>&1 | line "fuse   "
VPIC_HIP_SORT_IN_PUSH=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-config $cfg 2>&1 | line "no fuse"
done; done
