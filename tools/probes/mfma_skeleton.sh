#!/bin/bash
# Builds and times tools/probes/mfma_skeleton.hip, then its counters (rocprofv3 --pmc); writes gpurun_out/mfma_skeleton.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/probes/mfma_skeleton.hip -o /tmp/mfma_skel 2>/dev/null || exit 1
/tmp/mfma_skel 64 | tee $R/gpurun_out/mfma_skeleton.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/skel_pmc
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/skel_pmc -- /tmp/mfma_skel 64 > /tmp/skel_pmc.log 2>&1
python3 - <<'PY' | tee -a $R/gpurun_out/mfma_skeleton.txt
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("/tmp/skel_pmc/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "skeleton" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for d in per.values():
        for c, v in d.items():
            acc[c].append(v)
frames = 256 * 8 * 64 * 16
for c in sorted(acc):
    m = sum(acc[c]) / len(acc[c])
    print("%-32s %.6g   per frame %.2f" % (c, m, m / frames))
PY
tail -3 /tmp/skel_pmc.log
