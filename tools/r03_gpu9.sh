#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03i; mkdir -p $O
VPIC_HIP_LIB=$PWD/tools/ab/libt384.so timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tiles.py -m gpu -x -q > $O/pytest_t384.log 2>&1; echo "pytest(t384) rc=$?"; tail -3 $O/pytest_t384.log
echo "== config 2 (256^3, 64 ppc)"; bash tools/ab.sh "cur t384 t512 mq64" "" 2>&1 | tee $O/ab_c2.txt
echo "== config 1 (128^3, 32 ppc)"; bash tools/ab.sh "cur t384 t512 mq64" "--config 1" 2>&1 | tee $O/ab_c1.txt
echo "== config 2 deterministic"; bash tools/ab.sh "cur" "--accumulation deterministic" 2>&1 | tee $O/ab_c2_det.txt
echo "== config 1 deterministic"; bash tools/ab.sh "cur" "--config 1 --accumulation deterministic" 2>&1 | tee $O/ab_c1_det.txt
