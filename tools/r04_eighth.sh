#!/bin/bash
# the whole GPU suite once more, then the profile script
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04h; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/r04_profiles.sh
