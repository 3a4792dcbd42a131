#!/bin/bash
# the whole GPU suite and the smoke entry on the tree as it is
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03full; mkdir -p $O
timeout -k 10 1700 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 900 python tools/longrun_fuse.py 2>&1 | grep -v amdgpu.ids | tee $O/longrun_fuse.txt
