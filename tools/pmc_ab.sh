#!/bin/bash
# usage: tools/pmc_ab.sh <cfg> <lib or -> ...  -- LDS / issue counters per frame of the front-end kernel for each build
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CFG=$1; shift
for L in "$@"; do
  if [ "$L" = "-" ]; then unset CTU_ENGINE_LIB; else export CTU_ENGINE_LIB=$R/$L; fi
  T=$(echo $L | tr '/.' '__'); O=$R/gpurun_out/pmcab_$T; rm -rf $O; mkdir -p $O
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/a -- python3 $R/tools/bench_cfg.py --cfg $CFG --steps 2 > $O/log.txt 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $O/b -- python3 $R/tools/bench_cfg.py --cfg $CFG --steps 2 >> $O/log.txt 2>&1
  python3 - $O "$L" <<'PY'
import csv,glob,sys,collections,json
agg=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/**/*_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'frontend' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
frames=json.loads([l for l in open(sys.argv[1]+'/log.txt') if l.startswith('{')][-1])['frames']
print(sys.argv[2], 'frames', frames, {k:'%.4g'%(sum(v)/len(v)/frames) for k,v in sorted(agg.items())})
PY
done
