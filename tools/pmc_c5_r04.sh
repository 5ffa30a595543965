#!/bin/bash
# Round 4: what trapdct_split16_kernel waits for: wait / LDS / matrix-pipe / L2 write counters of the kernel at
# 10 000 utterances, three separate --pmc passes beside a kernel trace.  usage (GPU box): bash tools/pmc_c5_r04.sh [lib] > gpurun_out/pmc_c5_r04.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -n "$1" ] && export CTU_ENGINE_LIB=$R/$1
O=$R/gpurun_out/pmc_c5_${2:-r04}; rm -rf $O; mkdir -p $O
CMD="python3 $R/tools/bench_cfg.py --cfg C5 --utts 10000 --steps 2"
rocprofv3 -L 2>/dev/null | grep -o "TCC_[A-Z0-9_]*WR[A-Z0-9_]*\|SQ_[A-Z_]*BARRIER[A-Z_]*\|SQ_WAIT[A-Z_]*\|TCP_[A-Z_]*WRITE[A-Z_]*" | sort -u | tr '\n' ' ' > $O/counters_available.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- $CMD > $O/log.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/a -- $CMD >> $O/log.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/b -- $CMD >> $O/log.txt 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum GRBM_GUI_ACTIVE --output-format csv -d $O/c -- $CMD >> $O/log.txt 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SMEM --output-format csv -d $O/d -- $CMD >> $O/log.txt 2>&1
python3 - $O <<'PY'
import csv,glob,sys,collections
for pat in ("trapdct_split16", "frontend_kernel"):
    agg=collections.defaultdict(list)
    for f in glob.glob(sys.argv[1]+'/[abcd]/**/*_counter_collection.csv', recursive=True):
        per=collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if pat in r['Kernel_Name']: per[r['Dispatch_Id']][r['Counter_Name']]+=float(r['Counter_Value'])
        for d in per.values():
            for k,v in d.items(): agg[k].append(v)
    print("==", pat)
    for k in sorted(agg): print('%-34s %.6g' % (k, sum(agg[k])/len(agg[k])))
    for f in glob.glob(sys.argv[1]+'/t/**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r['Name']: print('avg ns', r['AverageNs'], 'calls', r['Calls'], r['Name'][:60])
print(open(sys.argv[1]+'/counters_available.txt').read())
PY
