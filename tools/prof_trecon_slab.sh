#!/bin/bash
# rocprofv3 kernel statistics of the production deck at its per-GPU slab size (oracle/_ref/treconslab.hip.exe, see time_trecon_slab.sh)
cd "$(dirname "$0")/.."; ulimit -c 0
OUT=$PWD/gpurun_out/trecon_slab_prof; rm -rf $OUT; mkdir -p $OUT; W=/tmp/trecon_slab_prof_run; rm -rf $W; mkdir -p $W; cd $W
export TMPDIR=/tmp
( while sleep 50; do echo "  ..."; done ) & KA=$!
VPIC_HIP_HOST_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d $W/prof -- /root/repo/oracle/_ref/treconslab.hip.exe -tpp=1 > log 2>&1 || true
kill $KA 2>/dev/null
grep -E "simulation time|hip host|rror" log
f=$(find $W/prof -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("%-70s %8s %12s %10s %6s" % ("kernel","calls","total_ms","avg_us","pct"))
for r in rows[:24]:
    print("%-70s %8s %12.3f %10.1f %6s" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
cd /; rm -rf $W
