# usage: tools/prof_stats.sh <tag> [bench args]  -> gpurun_out/<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --device-warmup-s 0 --no-cpu-baseline --no-second-config "$@" > gpurun_out/prof_$tag.log 2>&1
tail -2 gpurun_out/prof_$tag.log
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("%-60s %8s %12s %10s %6s" % ("kernel","calls","total_ms","avg_us","pct"))
for r in rows[:14]:
    print("%-60s %8s %12.3f %10.1f %6s" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
