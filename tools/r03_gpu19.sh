#!/bin/bash
# hot scalars of the main pass in vector registers (no v_readlane of spilled SGPRs): parity subset + A/B against HEAD
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03t; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tiles.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== config 2"; bash tools/ab.sh "prev cur" ""
echo "== config 1"; bash tools/ab.sh "prev cur" "--config 1"
