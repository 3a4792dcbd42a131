# usage: tools/pmc_push.sh <tag> [bench args]   -- PMC counters of advance_p_kernel (two passes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  c=$(echo $set | cut -d' ' -f1)
  rm -rf gpurun_out/pmc_${tag}_$c
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_${tag}_$c -- python3 bench.py --device-warmup-s 0 --no-cpu-baseline "$@" > gpurun_out/pmc_${tag}_$c.log 2>&1
done
python3 - $tag <<'PY'
import csv,glob,collections,sys
tag=sys.argv[1]
tot=collections.defaultdict(float); n=0
for f in sorted(glob.glob('gpurun_out/pmc_%s_*/**/*counter_collection.csv' % tag, recursive=True)):
    disp=set()
    for r in csv.DictReader(open(f)):
        if 'advance_p' in r['Kernel_Name']:
            tot[r['Counter_Name']]+=float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
    n=max(n,len(disp))
w=tot['SQ_WAVES']
print(tag,'launches',n,'waves/launch',w/n)
for c,x in sorted(tot.items()): print('   %-24s total %.4g   per wave %.1f' % (c,x,x/w))
PY
