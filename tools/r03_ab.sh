#!/bin/bash
# usage: tools/r03_ab.sh "<specs>" [<lib to run the kernel parity tests on>] -- parity tests on one variant, then A/B on the bench decks
cd "$(dirname "$0")/.."; ulimit -c 0
O=gpurun_out/r03ab; mkdir -p $O
if [ -n "$2" ]; then
  VPIC_HIP_LIB=$PWD/tools/ab/lib$2.so timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tiles.py -m gpu -x -q > $O/pytest_$2.log 2>&1; echo "pytest($2) rc=$?"; tail -4 $O/pytest_$2.log
fi
echo "== config 1 (128^3, 32 ppc)"; bash tools/ab.sh "$1" "--config 1" 2>&1 | tee $O/ab_c1.txt
echo "== config 2 (256^3, 64 ppc)"; bash tools/ab.sh "$1" "" 2>&1 | tee $O/ab_c2.txt
