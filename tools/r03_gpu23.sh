#!/bin/bash
# XCD-contiguous chunks for the sort's scatter: the scatter's time against the round-robin mapping
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03x; mkdir -p $O
VPIC_HIP_LIB=$PWD/tools/ab/libsxcd.so timeout -k 10 900 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest(sxcd) rc=$?"; tail -3 $O/pytest.log
echo "== cur"; bash tools/prof_stats.sh r03x_cur --steps 20 --warmup 5 | grep "kernel \|scatter\|count_kernel"
echo "== sxcd"; VPIC_HIP_LIB=$PWD/tools/ab/libsxcd.so bash tools/prof_stats.sh r03x_sxcd --steps 20 --warmup 5 | grep "kernel \|scatter\|count_kernel"
echo "== c1 cur"; bash tools/prof_stats.sh r03x_c1_cur --steps 20 --warmup 5 --config 1 | grep "kernel \|scatter\|count_kernel"
echo "== c1 sxcd"; VPIC_HIP_LIB=$PWD/tools/ab/libsxcd.so bash tools/prof_stats.sh r03x_c1_sxcd --steps 20 --warmup 5 --config 1 | grep "kernel \|scatter\|count_kernel"
