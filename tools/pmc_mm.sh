#!/bin/bash
# PMC pass over the phase-2-only run of the matrix-core variant
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_mm
mkdir -p $O
export CTU_DEBUG_MODE=${1:-2}
export CTU_DEBUG2=${2:-0}
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC --output-format csv -d $O/p1 -- python3 $R/tools/bench_cfg.py --cfg C2 --steps 2 > $O/p1.log 2>&1
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$O/p1/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
    for k in acc:
        if "frontend" in k:
            print(k, n[k]); 
            for c, v in acc[k].items(): print("   ", c, v / max(n[k], 1))
PY
tail -3 $O/p1.log
