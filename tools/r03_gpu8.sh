#!/bin/bash
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r03h; mkdir -p $O
timeout -k 10 1200 python -m pytest tests/test_domain_gloo.py tests/test_gpu_deterministic.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
timeout -k 10 600 python -m pytest tests/test_gpu_deck_host.py -m gpu -x -q -k "cleaning" > $O/pytest_deck.log 2>&1; echo "pytest deck rc=$?"; tail -12 $O/pytest_deck.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/overlap; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/overlap -- python3 tools/overlap_trace.py > $O/overlap_run.txt 2>&1; echo "overlap rc=$?"; grep "^domain" $O/overlap_run.txt
python3 tools/overlap_trace.py --analyse gpurun_out/overlap 2>&1 | tee $O/overlap_analysis.txt
find gpurun_out/overlap -name "*_trace.csv" -size +20M -delete
python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03h/bench.json') if l.startswith('{')][-1])
print('value %.2f G  ms/step %.2f (median %.2f)  frac %.4f  32ppc %.4f  fast %.4f  si20 %.2f G frac %.4f' % (d['value']/1e9, d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], d['roofline_32ppc']['frac'], d['roofline_fast']['frac'], d['same_deck_sort_interval_20']['value']/1e9, d['same_deck_sort_interval_20']['roofline']['frac']))
c=d['config3_slab']; print('config3 slab: value %.2f G  ms/step %.2f  push frac %.4f  avg launch %.3f ms' % (c['value']/1e9, c['ms_per_step'], c['roofline']['frac'], c['roofline']['avg_launch_ms']))
PY
