# usage: tools/pmc_traffic.sh <tag> [bench args] -- HBM traffic of advance_p_kernel from the TCC counters.
# FETCH_SIZE and WRITE_SIZE are collected in separate --pmc passes (MI355X_MICROARCH.md: TCC has 4
# slots, FETCH_SIZE takes 3, WRITE_SIZE 2), with --kernel-trace only.  Both are in KiB-like units of
# 1024 B?? -> calibrated below on kernels of known traffic and the same access width (dword per lane):
#   reads : sort_count_kernel      reads exactly 4 B per particle
#   writes: load_maxwellian_kernel writes exactly 32 B per particle
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_${tag}_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -- python3 bench.py --device-warmup-s 0 --no-cpu-baseline "$@" > gpurun_out/pmc_${tag}_$c.log 2>&1
done
python3 - $tag "$@" <<'PY'
import csv,glob,collections,sys,json
tag=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/pmc_%s_*/**/*counter_collection.csv' % tag, recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].split('::')[-1]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
out={}
for k,v in agg.items():
    out[k]={c:{'launches':len(x),'mean':sum(x)/len(x)} for c,x in v.items()}
    print(k, {c:(len(x), round(sum(x)/len(x),1)) for c,x in v.items()})
json.dump(out, open('gpurun_out/traffic_%s_raw.json' % tag,'w'), indent=1)
PY
