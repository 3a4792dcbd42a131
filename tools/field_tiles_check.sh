#!/bin/bash
# advance_b / advance_e through LDS tiles: parity tests, then the field kernels' times at 256^3 and 128^3 either way (rocprofv3 kernel stats of 12 steps)
cd "$(dirname "$0")/.."; ulimit -c 0; export VPIC_HIP_NO_REBUILD=1
O=gpurun_out/r04ft; mkdir -p $O
echo "(parity tests: see the full suite)"
cd /tmp; export TMPDIR=/tmp
for cfg in 2 1; do for ft in 2 0; do
  rm -rf /tmp/prof_ft; VPIC_HIP_FIELD_TILES=$ft timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ft -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-second-config --config $cfg --steps 10 --warmup 2 > /dev/null 2>$GRAFT_REPO_ROOT/$O/prof_${cfg}_${ft}.err
  f=$(find /tmp/prof_ft -name "*kernel_stats.csv" | head -1)
  echo "-- config $cfg FIELD_TILES=$ft"; grep -E "advance_b|advance_e|load_interpolator|clear_unload" "$f" | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin):
    print('   %-62s calls %5s avg %8.1f us' % (r[0][:62], r[1], float(r[3])/1e3))"
  cp "$f" $GRAFT_REPO_ROOT/$O/kernel_stats_config${cfg}_tiles${ft}.csv
done; done
