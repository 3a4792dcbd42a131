#!/bin/bash
# sort intervals with the window following its tile's particles up to 3 cells
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tiles.py -m gpu -x -q 2>&1 | tail -2
for si in 10 20 25 30; do python bench.py --no-cpu-baseline --no-second-config --sort-interval $si --steps $((2*si)) --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('sort_interval $si: %.2f G pushes/s  %.2f ms/step  advance_p %.2f ms/launch  roofline %.3f' % (j['value']/1e9, j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac']))"; done
for si in 10 20 30; do python bench.py --no-cpu-baseline --no-second-config --config 1 --sort-interval $si --steps $((2*si)) --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.readline()); print('config 1 sort_interval $si: %.2f G pushes/s  %.2f ms/step  advance_p %.3f ms/launch  roofline %.3f' % (j['value']/1e9, j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['frac']))"; done
