#!/bin/bash
# usage: tools/r02_ab2.sh "<specs>" [bench args]: A/B on the default deck (256^3 x 64 ppc) and 128^3 x 64 ppc; spec = lib[:mode][@env=val]
cd "$(dirname "$0")/.."; ulimit -c 0
for rep in 1 2; do for spec in $1; do
  envs=""; s=$spec
  if [[ "$s" == *@* ]]; then envs=${s#*@}; s=${s%%@*}; fi
  l=${s%%:*}; mode=exact; [ "$s" != "$l" ] && mode=${s##*:}
  if [ "$l" = cur ]; then unset VPIC_HIP_LIB; else export VPIC_HIP_LIB=$PWD/tools/ab/lib$l.so; fi
  echo -n "$spec: "; env ${envs:+$envs} python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-second-config --push $mode $2 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.2f G/s  push %.2f G/s  avg_launch %.3f ms  frac %.3f  ms/step %.2f' % (d['value']/1e9, d['advance_p_pushes_per_s']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step']))
    elif 'rror' in l: print(l.strip())"
done; done
