#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03r; mkdir -p $O
VPIC_HIP_LIB=$PWD/tools/ab/libst512.so timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tiles.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest(st512) rc=$?"; tail -3 $O/pytest.log
echo "== cur (256 threads)"; bash tools/prof_stats.sh r03r_cur --steps 20 --warmup 5 | grep "kernel \|scatter\|count_kernel"
echo "== st512"; VPIC_HIP_LIB=$PWD/tools/ab/libst512.so bash tools/prof_stats.sh r03r_512 --steps 20 --warmup 5 | grep "kernel \|scatter\|count_kernel"
echo "== hot cur"; bash tools/prof_stats.sh r03r_hot_cur --steps 20 --warmup 10 --config 1 --vth 0.6 --sort-interval -20 | grep "kernel \|scatter\|count_kernel"
echo "== hot st512"; VPIC_HIP_LIB=$PWD/tools/ab/libst512.so bash tools/prof_stats.sh r03r_hot_512 --steps 20 --warmup 10 --config 1 --vth 0.6 --sort-interval -20 | grep "kernel \|scatter\|count_kernel"
