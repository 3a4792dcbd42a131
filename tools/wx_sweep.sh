for rep in 1 2; do
 for cfg in "wx72 8" "wx62 7" "wx62 6" "wx72 7"; do set -- $cfg
  export VPIC_HIP_LIB=$PWD/tools/ab/lib$1.so VPIC_HIP_ITERS=$2
  echo -n "$1 iters=$2: "; python bench.py --steps 10 --warmup 3 --no-cpu-baseline $EXTRA 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.2f G/s  push %.2f G/s  avg_launch %.3f ms  ms/step %.2f' % (d['value']/1e9, d['advance_p_pushes_per_s']/1e9, d['roofline']['avg_launch_ms'], d['ms_per_step']))"
 done; done
