#!/bin/bash
# float instead of double sums in the tile window (no v_cvt_f64_f32, 10 KB less LDS): parity of the variant + A/B
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03u; mkdir -p $O
VPIC_HIP_LIB=$PWD/tools/ab/libfacc.so timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_tiles.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest(facc) rc=$?"; tail -5 $O/pytest.log
echo "== config 2"; bash tools/ab.sh "cur facc facc64" ""
echo "== config 1"; bash tools/ab.sh "cur facc facc64" "--config 1"
echo "== hot"; bash tools/ab.sh "cur facc" "--config 1 --vth 0.6 --sort-interval -20"
