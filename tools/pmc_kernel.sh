#!/bin/bash
# usage: tools/pmc_kernel.sh <cfg> <kernel substring>  -- counters of one kernel of a bench_cfg.py configuration, per launch
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=$1; PAT=$2
O=$R/gpurun_out/pmck_$CFG; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/bench_cfg.py --cfg $CFG --steps 3 > $O/log.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/a -- python3 $R/tools/bench_cfg.py --cfg $CFG --steps 2 >> $O/log.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/b -- python3 $R/tools/bench_cfg.py --cfg $CFG --steps 2 >> $O/log.txt 2>&1
# (a third pass with FETCH_SIZE / WRITE_SIZE / GRBM_GUI_ACTIVE hung twice on the multi-kernel C5 run and is left out)
python3 - $O "$PAT" <<'PY'
import csv,glob,sys,collections
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/[ab]/**/*_counter_collection.csv', recursive=True):
    per=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']: per[r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value'])
    for d in per.values():
        for k,v in d.items(): agg[k].append(v)
for k in sorted(agg): print('%-34s %.6g' % (k, sum(agg[k])/len(agg[k])))
for f in glob.glob(sys.argv[1]+'/t/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Name']: print('avg ns', r['AverageNs'], 'calls', r['Calls'])
PY
