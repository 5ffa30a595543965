#!/bin/bash
# usage: tools/prof.sh <tag> [bench args]  -- kernel trace + PMC passes of a short bench run, outputs under gpurun_out/prof_<tag>
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu ${@:---utts 2000}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $BENCH > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/pmc1 -- $BENCH > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $O/pmc2 -- $BENCH > $O/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc3 -- $BENCH > $O/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/pmc4 -- $BENCH > $O/pmc4.log 2>&1 || true
python3 - $O <<'PY'
import csv, glob, sys, collections, os
O = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(O + '/pmc*/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'frontend' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
with open(O + '/pmc_summary.txt', 'w') as out:
    for k, v in sorted(agg.items()):
        out.write('%-28s %.6g   (mean over %d dispatches)\n' % (k, sum(v) / len(v), len(v)))
for f in glob.glob(O + '/trace/**/*kernel_stats.csv', recursive=True):
    os.system('cp %s %s/kernel_stats.csv' % (f, O))
print(open(O + '/pmc_summary.txt').read())
PY
tail -2 $O/trace.log | cut -c1-600
