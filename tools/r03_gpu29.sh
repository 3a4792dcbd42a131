#!/bin/bash
# the sort inside the push: its test, the tile and kernel tests, then the default deck with and without it
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_tiles.py -m gpu -x -q -k "inside_the_push" > $O/pytest1.log 2>&1; echo "pytest(fuse) rc=$?"; tail -15 $O/pytest1.log
timeout -k 10 900 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest2.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest2.log
echo "== config 1: fuse / no fuse"
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-config --config 1 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse   : value %.2f G/s ms/step %.2f avg_launch %.3f frac %.3f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
VPIC_HIP_NO_FUSE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-second-config --config 1 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('no fuse: value %.2f G/s ms/step %.2f avg_launch %.3f frac %.3f' % (d['value']/1e9, d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
