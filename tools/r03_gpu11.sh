#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03k; mkdir -p $O
timeout -k 10 1200 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_kernels.py tests/test_gpu_tiles.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
echo "== sort"; bash tools/prof_stats.sh r03k_new --steps 10 --warmup 3 | grep "kernel \|scatter\|count_kernel\|advance_p"
