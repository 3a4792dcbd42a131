#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03l; mkdir -p $O
echo "== trecon slab"; bash tools/prof_stats.sh r03l_trecon --steps 10 --warmup 8 --deck trecon --sort-interval -20
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03l/bench.json') if l.startswith('{')][-1])
print('value %.2f G  ms/step %.2f (median %.2f)  frac %.4f  32ppc %.4f  fast %.4f  si20 %.2f G frac %.4f' % (d['value']/1e9, d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], d['roofline_32ppc']['frac'], d['roofline_fast']['frac'], d['same_deck_sort_interval_20']['value']/1e9, d['same_deck_sort_interval_20']['roofline']['frac']))
c=d['config3_slab']; print('config3 slab: value %.2f G  ms/step %.2f  push frac %.4f  avg launch %.3f ms' % (c['value']/1e9, c['ms_per_step'], c['roofline']['frac'], c['roofline']['avg_launch_ms']))
PY
