#!/bin/bash
# The GPU calls of round 3 as they were run: gpurun --timeout 1200 -- 'bash tools/gpu_round3.sh <step>'.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
case "$1" in
  probes)   # ahead of the matrix-pipe decision and for the tolerance items of VERDICT r02
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_acc.hip -o /tmp/mfma_acc 2>/dev/null && /tmp/mfma_acc > gpurun_out/mfma_acc.txt 2>&1
    python tools/probes/exten_err.py C2 > gpurun_out/c2_err.txt 2>&1
    python tools/probes/ss_vad_err.py > gpurun_out/ss_vad_err.txt 2>&1
    python tools/probes/ss16_dbg.py > gpurun_out/ss16_dbg.txt 2>&1
    tail -5 gpurun_out/mfma_acc.txt gpurun_out/c2_err.txt gpurun_out/ss_vad_err.txt gpurun_out/ss16_dbg.txt ;;
  tests)
    timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/gpu_tests.log ;;
  drec)     # the Burg lattice's denominator by its recursion (lib built with -DCTU_BURG_DREC=1 into ctucopy_amd/_variants/)
    for L in - ctucopy_amd/_variants/lib_drec.so; do
      if [ "$L" = "-" ]; then unset CTU_ENGINE_LIB; else export CTU_ENGINE_LIB=$L; fi
      echo "== $L"; python tools/bench_cfg.py --cfg C4_10k --steps 5 | tail -1; python tools/probes/exten_err.py C4 | tail -1
    done 2>&1 | tee gpurun_out/drec_ab.txt
    CTU_ENGINE_LIB=ctucopy_amd/_variants/lib_drec.so timeout -k 10 600 python -m pytest tests -m gpu -q -k "c4 or fixture or vad or spectral" > gpurun_out/drec_tests.log 2>&1
    tail -5 gpurun_out/drec_tests.log ;;
  cfgs)     # configurations touched this round, 10 000 utterances each
    for c in C2 C3 lp_noinld C4_10k C5 C2_vad16; do python tools/bench_cfg.py --cfg $c --utts 10000 --steps 5 | tail -1; done 2>&1 | tee gpurun_out/cfgs_r03.txt
    python tools/bench_cfg.py --cfg fft1024 --utts 2000 --steps 5 | tail -1 | tee -a gpurun_out/cfgs_r03.txt
    CTU_WAVE1K=0 python tools/bench_cfg.py --cfg fft1024 --utts 2000 --steps 5 | tail -1 | tee -a gpurun_out/cfgs_r03.txt ;;
  final)
    timeout -k 10 1500 bash tools/final_prof.sh r03 > gpurun_out/final_r03.log 2>&1; echo "final_prof rc $?"; tail -40 gpurun_out/final_r03.log ;;
  *) echo "usage: $0 probes|tests|drec|cfgs|final"; exit 2 ;;
esac
