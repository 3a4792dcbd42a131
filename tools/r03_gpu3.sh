#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03c; mkdir -p $O
./tools/ubench/late_stores.exe 2>&1 | tee $O/late_stores.txt
./tools/ubench/late_stores.exe 1073741824 2>&1 | tee -a $O/late_stores.txt
timeout -k 10 900 python -m pytest tests/test_domain_gloo.py -m gpu -x -q -k "grows" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
