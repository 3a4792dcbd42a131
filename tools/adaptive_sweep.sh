for deck in two-stream sheet; do for si in 10 5 -20; do echo -n "$deck sort_interval=$si: "; python bench.py --deck $deck --ppc 32 --steps 40 --warmup 10 --no-cpu-baseline --sort-interval $si 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.2f G/s  push %.2f G/s  avg_launch %.3f ms  ms/step %.2f' % (d['value']/1e9, d['advance_p_pushes_per_s']/1e9, d['roofline']['avg_launch_ms'], d['ms_per_step']))"; done; done
