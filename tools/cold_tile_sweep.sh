cd /root/repo; ulimit -c 0
for w in narrow tile; do for si in 5 10 20; do
  echo -n "cold c1 window=$w sort_interval=$si: "
  VPIC_HIP_WINDOW=$w python bench.py --config 1 --sort-interval $si --steps 40 --warmup 20 --no-cpu-baseline 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.2f G/s  avg_launch %.3f ms  frac %.3f  ms/step %.2f' % (d['value']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step']))
    elif 'rror' in l: print(l.strip())"
done; done
