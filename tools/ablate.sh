# usage: tools/ablate.sh "0 1 2 3" [extra bench args]   -- needs a library built with -DVPIC_HIP_ABLATION (VPIC_HIP_LIB, tools/build_wt_variant.sh)
for a in $1; do echo -n "ABLATE=$a: "; VPIC_HIP_ABLATE=$a python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-config $2 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.2f G/s  push %.2f G/s  avg_launch %.3f ms  ms/step %.2f' % (d['value']/1e9, d['advance_p_pushes_per_s']/1e9, d['roofline']['avg_launch_ms'], d['ms_per_step']))
    elif 'rror' in l: print(l.strip())"; done
