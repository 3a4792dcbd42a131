#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 1200 python -m pytest tests/test_gpu_deterministic.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_tiles.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout -k 10 600 python -m pytest tests/test_gpu_deck_host.py -m gpu -x -q -k "cleaning" > $O/pytest_deck.log 2>&1; echo "pytest deck rc=$?"; tail -8 $O/pytest_deck.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/overlap; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/overlap -- python3 tools/overlap_trace.py > $O/overlap_run.txt 2>&1; echo "overlap rc=$?"; tail -3 $O/overlap_run.txt
python3 tools/overlap_trace.py --analyse gpurun_out/overlap 2>&1 | tee $O/overlap_analysis.txt
find gpurun_out/overlap -name "*_trace.csv" -size +20M -delete
