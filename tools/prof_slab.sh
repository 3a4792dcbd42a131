# usage: tools/prof_slab.sh <tag>  -- rocprofv3 kernel stats of ONE x-slab of BASELINE configs[2] (32 x 256 x 256 cells,
# 2 species x 64 ppc: what each of 8 GPUs holds), field kernels included
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1
rm -rf gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --no-cpu-baseline --no-second-config --grid 32 256 256 --ppc 64 --steps 20 --warmup 5 > gpurun_out/prof_$tag.log 2>&1
tail -1 gpurun_out/prof_$tag.log | cut -c1-400
f=$(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("%-60s %8s %12s %10s %6s" % ("kernel","calls","total_ms","avg_us","pct"))
for r in rows[:16]:
    print("%-60s %8s %12.3f %10.1f %6s" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
rm -rf gpurun_out/prof_$tag
