#!/bin/bash
# usage: tools/pmc_modes.sh "<modes>" -- instruction-count PMC pass per CTU_DEBUG_MODE
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in ${1:-0 1}; do
  O=$R/gpurun_out/pmcm_$m; rm -rf $O; mkdir -p $O
  CTU_DEBUG_MODE=$m rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O -- python3 $R/bench.py --steps 2 --warmup 1 --utts 2000 --no-cpu > $O/log.txt 2>&1
  python3 - $O $m <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'frontend' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print('mode',sys.argv[2],{k:'%.4g'%(sum(v)/len(v)) for k,v in sorted(agg.items())})
PY
done
