#!/bin/bash
# the sort by tile only through the staged workgroup scatter: parity subset, hot decks against HEAD
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_tiles.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== hot vth 0.6"; bash tools/ab.sh "prev cur" "--config 1 --vth 0.6 --sort-interval -20 --steps 40 --warmup 10"
echo "== trecon"; bash tools/ab.sh "prev cur" "--deck trecon --sort-interval -20 --steps 40 --warmup 10"
echo "== sheet"; bash tools/ab.sh "prev cur" "--deck sheet --sort-interval -20 --steps 40 --warmup 10"
echo "== hot stats"; bash tools/prof_stats.sh r04a_hot --steps 40 --warmup 10 --config 1 --vth 0.6 --sort-interval -20 | grep "kernel \|scatter\|count_kernel"
