#!/bin/bash
# scatter with two staging buffers and prefetched loads: parity subset, then the scatter's time against HEAD's
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03v; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tiles.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== cur"; bash tools/prof_stats.sh r03v_cur --steps 20 --warmup 5 | grep "kernel \|scatter\|count_kernel"
echo "== prev"; VPIC_HIP_LIB=$PWD/tools/ab/libprev.so bash tools/prof_stats.sh r03v_prev --steps 20 --warmup 5 | grep "kernel \|scatter\|count_kernel"
echo "== hot cur"; bash tools/prof_stats.sh r03v_hot_cur --steps 20 --warmup 10 --config 1 --vth 0.6 --sort-interval -20 | grep "kernel \|scatter\|count_kernel"
echo "== hot prev"; VPIC_HIP_LIB=$PWD/tools/ab/libprev.so bash tools/prof_stats.sh r03v_hot_prev --steps 20 --warmup 10 --config 1 --vth 0.6 --sort-interval -20 | grep "kernel \|scatter\|count_kernel"
