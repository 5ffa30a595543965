#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
python tools/probes/ss16_dbg.py > gpurun_out/ss16_dbg.txt 2>&1; tail -40 gpurun_out/ss16_dbg.txt
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/gpu_tests.log
for L in - ctucopy_amd/_variants/lib_drec.so; do
  if [ "$L" = "-" ]; then unset CTU_ENGINE_LIB; else export CTU_ENGINE_LIB=$L; fi
  echo "== $L"; python tools/bench_cfg.py --cfg C4_10k --steps 5 | tail -1
  python tools/probes/exten_err.py C4 | tail -1
done 2>&1 | tee gpurun_out/drec_ab.txt
unset CTU_ENGINE_LIB
CTU_ENGINE_LIB=ctucopy_amd/_variants/lib_drec.so timeout -k 10 600 python -m pytest tests -m gpu -q -k "c4 or fixture or vad or spectral" > gpurun_out/drec_tests.log 2>&1; echo "drec pytest rc $?"; tail -5 gpurun_out/drec_tests.log
