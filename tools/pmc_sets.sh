# usage: [KERNEL=<substring>] tools/pmc_sets.sh <tag> "<set1>;<set2>;..." [bench args]   -- PMC counters of one kernel (default advance_p), one pass per set
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; sets=$2; shift; shift
IFS=';' read -ra SETS <<< "$sets"
k=0
for set in "${SETS[@]}"; do
  k=$((k+1))
  rm -rf gpurun_out/pmc_${tag}_$k
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$k -- python3 bench.py --device-warmup-s 0 --no-cpu-baseline --no-second-config "$@" > gpurun_out/pmc_${tag}_$k.log 2>&1
done
python3 - $tag "${KERNEL:-advance_p}" <<'PY'
import csv,glob,collections,sys
tag=sys.argv[1]; kern=sys.argv[2]
tot=collections.defaultdict(float); n=0
for f in sorted(glob.glob('gpurun_out/pmc_%s_*/**/*counter_collection.csv' % tag, recursive=True)):
    disp=set()
    for r in csv.DictReader(open(f)):
        if kern in r['Kernel_Name']:
            tot[r['Counter_Name']]+=float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
    n=max(n,len(disp))
print(tag,'launches',n)
for c,x in sorted(tot.items()): print('   %-32s per launch %.5g' % (c,x/n))
PY
rm -rf gpurun_out/pmc_${tag}_[0-9]*
