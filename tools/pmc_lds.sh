#!/bin/bash
# usage: tools/pmc_lds.sh "<modes>" -- LDS counters per CTU_DEBUG_MODE
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in ${1:-0 1 2}; do
  O=$R/gpurun_out/pmcl_$m; rm -rf $O; mkdir -p $O
  CTU_DEBUG_MODE=$m rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O -- python3 $R/bench.py --steps 2 --warmup 1 --utts 2000 --no-cpu > $O/log.txt 2>&1
  python3 - $O $m <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'frontend' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print('mode',sys.argv[2],{k:'%.4g'%(sum(v)/len(v)) for k,v in sorted(agg.items())})
PY
done
