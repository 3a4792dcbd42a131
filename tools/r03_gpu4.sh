#!/bin/bash
# full GPU suite on the committed tree, the bench line, and the 2-rank gloo rehearsal of the N > 1 bench path
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03d; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03d/bench.json') if l.startswith('{')][-1])
print('value %.2f G  ms/step %.2f (median %.2f)  frac %.4f  32ppc %.4f  fast %.4f  si20 %.2f G frac %.4f' % (d['value']/1e9, d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], d['roofline_32ppc']['frac'], d['roofline_fast']['frac'], d['same_deck_sort_interval_20']['value']/1e9, d['same_deck_sort_interval_20']['roofline']['frac']))
PY
VPIC_HIP_SINGLE_DEVICE=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --backend gloo --grid 128 128 128 > $O/bench2.json 2> $O/bench2.err; echo "bench2 rc=$?"; tail -c 1500 $O/bench2.json
