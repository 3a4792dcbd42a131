#!/bin/bash
# round 2, first GPU call: new parity tests, then A/B of the push variants on one box
cd "$(dirname "$0")/.."; ulimit -c 0
O=gpurun_out/r02a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log
echo "== config 1 (128^3, 32 ppc)"; bash tools/ab.sh "r01 cur cur:fast" "--config 1" 2>&1 | tee $O/ab_c1.txt
echo "== 128^3, 64 ppc"; bash tools/ab.sh "r01 cur cur:fast" "--config 1 --ppc 64" 2>&1 | tee $O/ab_c1_64.txt
echo "== config 2 (256^3, 64 ppc)"; bash tools/ab.sh "cur cur:fast" "" 2>&1 | tee $O/ab_c2.txt
echo "== hot vth 0.6 adaptive"; bash tools/ab.sh "r01 cur cur:fast" "--config 1 --vth 0.6 --sort-interval -20" 2>&1 | tee $O/ab_hot.txt
