#!/bin/bash
# usage: tools/r02_ab.sh "<specs>" -- kernel parity tests on the current build, then A/B on the bench decks
cd "$(dirname "$0")/.."; ulimit -c 0
O=gpurun_out/r02b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
echo "== config 1 (128^3, 32 ppc)"; bash tools/ab.sh "$1" "--config 1" 2>&1 | tee $O/ab_c1.txt
echo "== 128^3, 64 ppc"; bash tools/ab.sh "$1" "--config 1 --ppc 64" 2>&1 | tee $O/ab_c1_64.txt
echo "== hot vth 0.6 adaptive"; bash tools/ab.sh "$1" "--config 1 --vth 0.6 --sort-interval -20" 2>&1 | tee $O/ab_hot.txt
