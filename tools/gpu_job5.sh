#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/gpu_tests.log
for c in C3 lp_noinld C4_10k C4 C2_vad16; do python tools/bench_cfg.py --cfg $c --utts 10000 --steps 5 | tail -1; done 2>&1 | tee gpurun_out/cfgs_r03b.txt
python tools/probes/exten_err.py C4 | tail -1
