#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q -k "fft_sizes or lp_orders or odd_frame" > gpurun_out/gpu_tests_k.log 2>&1; echo "pytest rc $?"; tail -12 gpurun_out/gpu_tests_k.log
python tools/bench_cfg.py --cfg fft1024 --utts 2000 --steps 5 | tail -1
CTU_WAVE1K=0 python tools/bench_cfg.py --cfg fft1024 --utts 2000 --steps 5 | tail -1
