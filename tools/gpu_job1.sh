#!/bin/bash
# first GPU call of round 3: probes ahead of the matrix-pipe front end + the round's parity fixes + the new bench line
set -o pipefail
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_acc.hip -o /tmp/mfma_acc 2>/dev/null && /tmp/mfma_acc > gpurun_out/mfma_acc.txt 2>&1
cat gpurun_out/mfma_acc.txt
python tools/probes/exten_err.py C2 > gpurun_out/c2_err.txt 2>&1; cat gpurun_out/c2_err.txt
python tools/probes/ss_vad_err.py > gpurun_out/ss_vad_err.txt 2>&1; tail -20 gpurun_out/ss_vad_err.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/gpu_tests.log
timeout -k 10 600 python bench.py > gpurun_out/bench_r03a.json 2> gpurun_out/bench_r03a.err; echo "bench rc $?"; tail -c 3000 gpurun_out/bench_r03a.json; tail -5 gpurun_out/bench_r03a.err
