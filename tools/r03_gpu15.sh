#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tiles.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for rep in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export VPIC_HIP_NO_SORT_OVERLAP=1; else unset VPIC_HIP_NO_SORT_OVERLAP; fi
  echo -n "no_overlap=$v: "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-config 2>&1 | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value %.2f G/s  push %.2f G/s  avg_launch %.3f ms  frac %.4f  ms/step %.2f median %.2f' % (d['value']/1e9, d['advance_p_pushes_per_s']/1e9, d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['ms_per_step'], d['ms_per_step_median']))
    elif 'rror' in l: print(l.strip())"
done; done
