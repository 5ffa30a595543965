#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -15 gpurun_out/gpu_tests.log
python tools/bench_cfg.py --cfg C2_vad16 --utts 2000 > gpurun_out/c2_vad16.txt 2>&1; tail -2 gpurun_out/c2_vad16.txt
