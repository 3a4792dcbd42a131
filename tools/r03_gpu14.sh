#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03n; mkdir -p $O
timeout -k 10 1200 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
echo "== config 2 (256^3, 64 ppc)"; bash tools/ab.sh "head cur" "" 2>&1 | tee $O/ab_c2.txt
echo "== config 1 (128^3, 32 ppc)"; bash tools/ab.sh "head cur" "--config 1" 2>&1 | tee $O/ab_c1.txt
echo "== stats"; bash tools/prof_stats.sh r03n_cur --steps 20 --warmup 5 | grep "kernel \|scatter\|count_kernel\|advance_p\|tile_max"
