#!/bin/bash
# Round artifacts: the default bench line, the rocprofv3 kernel trace of the same workload, and the PMC passes.
# usage (on the GPU box): bash tools/final_prof.sh <tag>      outputs under gpurun_out/final_<tag>/
set -e
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_$TAG
mkdir -p $O
cd $R
python3 bench.py > $O/bench_full.json 2> $O/bench_full.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --no-cpu --no-extra > $O/trace.log 2>&1
P="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extra"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $O/pmc1 -- $P > $O/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $O/pmc2 -- $P > $O/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc3 -- $P > $O/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $O/pmc4 -- $P > $O/pmc4.log 2>&1
cd $R
python3 - "$O" <<'PY'
import csv, glob, collections, sys, json, os
O = sys.argv[1]
stats = glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    rows = list(csv.reader(open(stats[0])))
    with open(O + "/kernel_stats.csv", "w") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        for r in rows[:12]:
            w.writerow(r)
acc = collections.defaultdict(list)
for f in glob.glob(O + "/pmc*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "frontend_kernel" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for d in per.values():
        for c, v in d.items():
            acc[c].append(v)
with open(O + "/pmc_summary.txt", "w") as f:
    for c in sorted(acc):
        f.write("%-24s %.6g   (mean over %d dispatches)\n" % (c, sum(acc[c]) / len(acc[c]), len(acc[c])))
print(open(O + "/pmc_summary.txt").read())
print(open(O + "/bench_full.json").read()[:1500])
PY
