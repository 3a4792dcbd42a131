#!/bin/bash
# the default bench line with its riders in fresh processes behind the main deck
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
for rep in 1 2; do
S=$SECONDS; python bench.py --no-cpu-baseline 2> gpurun_out/r04k_err.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('default line: value %.2f G  %.2f ms/step  frac %.3f | c1 %.2f G frac %.3f | c3 %.2f G frac %.3f | si20 %.2f G frac %.3f | fast %.2f G frac %.3f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['frac'], d['config1_128cubed_32ppc']['value']/1e9, d['roofline_32ppc']['frac'], d['config3_slab']['value']/1e9, d['roofline_config3_slab']['frac'], d['same_deck_sort_interval_20']['value']/1e9, d['same_deck_sort_interval_20']['roofline']['frac'], d['same_deck_fast_arithmetic']['value']/1e9, d['roofline_fast']['frac']))"
echo "wall $((SECONDS-S)) s"; tail -1 gpurun_out/r04k_err.txt
python bench.py --no-cpu-baseline --no-second-config 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('alone       : value %.2f G  %.2f ms/step  frac %.3f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['frac']))"
done
