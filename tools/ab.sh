# usage: tools/ab.sh "<libA> <libB> ..." [bench args]   -- same bench against several builds, interleaved twice
# a name of the form <lib>:fast runs that build with --push fast; "cur" is old-vpic_amd/libvpic_hip.so
for rep in 1 2; do for spec in $1; do
  l=${spec%%:*}; mode=exact; [ "$spec" != "$l" ] && mode=${spec##*:}
  if [ "$l" = cur ]; then unset VPIC_HIP_LIB; else export VPIC_HIP_LIB=$PWD/tools/ab/lib$l.so; fi
  echo -n "$spec: "; python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-config --push $mode $2 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.2f G/s  push %.2f G/s  avg_launch %.3f ms  frac %.3f  ms/step %.2f  sorting launch %.2f ms' % (d['value']/1e9, d['advance_p_pushes_per_s']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step'], (d.get('advance_p_sorting') or {}).get('avg_launch_ms', 0)))
    elif 'rror' in l: print(l.strip())"
done; done
