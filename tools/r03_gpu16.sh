#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03q; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tiles.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== config 2 (256^3, 64 ppc), 20 steps"; bash tools/ab.sh "head cur" "--steps 20 --warmup 5" 2>&1 | tee $O/ab_c2.txt
echo "== stats"; bash tools/prof_stats.sh r03q_cur --steps 20 --warmup 5 | grep "kernel \|scatter\|count_kernel\|advance_p"
