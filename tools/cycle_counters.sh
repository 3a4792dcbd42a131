# usage: tools/cycle_counters.sh "<counters>" [bench args]: one PMC pass; the counters of every advance_p dispatch in launch order
# (how the kernel's work changes from the first step after a sort to the last)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
c=$1; shift
rm -rf gpurun_out/pmc_cycle
rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_cycle -- python3 bench.py --no-cpu-baseline --no-second-config "$@" > gpurun_out/pmc_cycle.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_cycle/**/*counter_collection.csv', recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'advance_p' in r['Kernel_Name']]
by=collections.OrderedDict()
for r in rows: by.setdefault(int(r['Dispatch_Id']),{})[r['Counter_Name']]=float(r['Counter_Value'])
names=sorted({k for v in by.values() for k in v})
print('dispatch', *names)
for d,v in by.items(): print(d, *['%.4g' % v.get(n,0) for n in names])
PY
rm -rf gpurun_out/pmc_cycle
