#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/gpu_tests.log
timeout -k 10 1500 bash tools/final_prof.sh r03 > gpurun_out/final_r03.log 2>&1; echo "final_prof rc $?"; tail -40 gpurun_out/final_r03.log
