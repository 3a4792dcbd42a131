#!/bin/bash
# the tile window follows the tile's particles (VPIC_HIP_FOLLOW, push.hip): parity subset, then sort intervals 10 / 20 / 30 against the fixed window
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04j; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tiles.py tests/test_gpu_fullsize.py tests/test_gpu_deterministic.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== config 2, interval 10"; bash tools/ab.sh "nofollow cur" "--steps 20 --warmup 5"
echo "== config 2, interval 20"; bash tools/ab.sh "nofollow cur" "--steps 40 --warmup 5 --sort-interval 20"
echo "== config 1, interval 10"; bash tools/ab.sh "nofollow cur" "--config 1 --steps 20 --warmup 5"
echo "== config 1, interval 20"; bash tools/ab.sh "nofollow cur" "--config 1 --steps 40 --warmup 5 --sort-interval 20"
