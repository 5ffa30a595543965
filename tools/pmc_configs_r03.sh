#!/bin/bash
# Instruction and LDS counters of the dominant kernels of the secondary configurations (tools/pmc_kernel.sh per configuration), per launch
# and per frame.  usage (GPU box): bash tools/pmc_configs_r03.sh > gpurun_out/pmc_configs_r03.txt
R=$GRAFT_REPO_ROOT
for pair in "C2:frontend_kernel" "C3:frontend_kernel" "C3:lp_tail" "C4:frontend_kernel" "C4:vad_lanes" "C5:trapdct_split16" "C2_vad16:frontend_kernel" "fft1024:wave1k" "exten_raw:frontend_kernel"; do
  c=${pair%%:*}; k=${pair##*:}
  echo "== $c / $k"
  bash $R/tools/pmc_kernel.sh $c $k 2>&1 | grep -v "^$" | tail -24
  grep -o '"frames": [0-9]*' $R/gpurun_out/pmck_$c/log.txt | head -1
done
