#!/bin/bash
# banded (XCD-aware) traversal of the per-step field kernels: parity subset, then kernel stats of the default bench and configs[1]
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03s; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_deck_host.py tests/test_gpu_dropin.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
echo "== banded fields, config 2"; bash tools/prof_stats.sh r03s_c2 --steps 10 --warmup 3 | grep -v "advance_p\|scatter\|maxwell"
echo "== banded fields, config 1"; bash tools/prof_stats.sh r03s_c1 --steps 10 --warmup 3 --config 1 | grep -v "advance_p\|scatter\|maxwell"
