#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_domain_gloo.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
export VPIC_HIP_LIB=$PWD/tools/ab/libabl.so
echo "== ablation config 2"; bash tools/ablate.sh "0 256 512 192 130 384" "" 2>&1 | tee $O/abl_c2.txt
echo "== ablation config 1"; bash tools/ablate.sh "0 256 512" "--config 1" 2>&1 | tee $O/abl_c1.txt
