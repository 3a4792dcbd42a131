for si in 2 3 5 10; do echo -n "sheet sort_interval=$si: "; python bench.py --deck sheet --ppc 32 --steps 20 --warmup 5 --no-cpu-baseline --sort-interval $si 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.2f G/s  push %.2f G/s  avg_launch %.3f ms  ms/step %.2f' % (d['value']/1e9, d['advance_p_pushes_per_s']/1e9, d['roofline']['avg_launch_ms'], d['ms_per_step']))"; done
