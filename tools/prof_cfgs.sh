#!/bin/bash
# rocprofv3 kernel stats of the engine's kernels for every configuration tools/bench_cfg.py knows (2000 utterances each)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/cfgs
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/summary.csv
echo '"config","kernel","calls","avg_ns","min_ns","max_ns"' >> $O/summary.csv
for c in C2 C3 C4_novad C4 C4_10k C5 C2_vad16 C2_fwss16 C2_d_a C2_trap9 C2_cms_exp C2_cms_block exten_raw lp_noinld C2_dc1 fft128 fft1024 fft1024_exten C2_fwss16_E_d_a; do
  rm -rf $O/t_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/t_$c -- python3 $R/tools/bench_cfg.py --cfg $c > $O/$c.log 2>&1 || { echo "failed $c"; exit 1; }
  f=$(find $O/t_$c -name "*kernel_stats.csv" | head -1)
  python3 - "$c" "$f" >> $O/summary.csv <<'PY'
import csv, sys
cfg, f = sys.argv[1], sys.argv[2]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("frontend_kernel", "post_kernel", "cms_", "vad_", "trapdct", "synth_kernel", "ola_kernel", "cmvn_", "bigfft", "dc1_", "lp_tail", "wave1k")):
        short = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        print('"%s","%s","%s","%.0f","%s","%s"' % (cfg, short, r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
  tail -1 $O/$c.log | cut -c1-160
done
cat $O/summary.csv
