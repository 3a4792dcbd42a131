#!/bin/bash
# the workgroup-level count / scatter against the wavefront-level one: parity tests, then kernel stats of both on the default deck
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03e; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tiles.py tests/test_gpu_fullsize.py tests/test_gpu_dropin.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
echo "== new sort"; bash tools/prof_stats.sh r03e_new --steps 10 --warmup 3
echo "== old sort"; VPIC_HIP_OLD_SORT=1 bash tools/prof_stats.sh r03e_old --steps 10 --warmup 3
echo "== new sort, config 1 hot"; bash tools/prof_stats.sh r03e_new_hot --steps 10 --warmup 3 --config 1 --vth 0.6 --sort-interval -20
echo "== old sort, config 1 hot"; VPIC_HIP_OLD_SORT=1 bash tools/prof_stats.sh r03e_old_hot --steps 10 --warmup 3 --config 1 --vth 0.6 --sort-interval -20
