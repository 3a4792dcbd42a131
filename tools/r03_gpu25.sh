#!/bin/bash
# chargeless species grouped by tile only: configs[3] slab + sheet deck against HEAD~ (prev), parity subset
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== trecon"; bash tools/ab.sh "prev cur" "--deck trecon --sort-interval -20 --steps 40 --warmup 10"
echo "== trecon stats"; bash tools/prof_stats.sh r03z_tr --steps 40 --warmup 10 --deck trecon --sort-interval -20 | grep "kernel \|scatter\|count_kernel\|advance_p"
