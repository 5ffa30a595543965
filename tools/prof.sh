#!/bin/bash
# usage: tools_prof.sh <tag>   -- kernel trace + three PMC passes of a short bench run, outputs under gpurun_out/prof_<tag>
set -e
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --utts 2000 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $BENCH > $O/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/pmc1 -- $BENCH > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $O/pmc2 -- $BENCH > $O/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc3 -- $BENCH > $O/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc4 -- $BENCH > $O/pmc4.log 2>&1
find $O -name "*.csv" | head -30
