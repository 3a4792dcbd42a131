#!/bin/bash
# the crossers' no-hit test in float against 2 d instead of in double against (2 - 2^-24) d: parity subset, hot and cold decks against HEAD
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tiles.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -2
echo "== hot vth 0.6"; bash tools/ab.sh "prev cur" "--config 1 --vth 0.6 --sort-interval -20 --steps 40 --warmup 10"
echo "== trecon"; bash tools/ab.sh "prev cur" "--deck trecon --sort-interval -20 --steps 40 --warmup 10"
echo "== config 2"; bash tools/ab.sh "prev cur" "--steps 20 --warmup 5"
echo "== config 1"; bash tools/ab.sh "prev cur" "--config 1 --steps 20 --warmup 5"
