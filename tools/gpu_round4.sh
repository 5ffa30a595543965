#!/bin/bash
# The GPU calls of round 4 as they were run: gpurun --timeout N -- 'bash tools/gpu_round4.sh <step> [args]'.  Outputs under gpurun_out/.
set -o pipefail
mkdir -p gpurun_out
V=ctucopy_amd/_variants
case "$1" in
  ab)       # ab <out> <rounds> <lib tags...>: interleaved A/B of engine builds on the headline workload, then the headline parity tests per build
    out=gpurun_out/$2; rounds=$3; shift 3
    libs=""; for t in "$@"; do if [ "$t" = "-" ]; then libs="$libs -"; else libs="$libs $V/lib_$t.so"; fi; done
    bash tools/ab_bench.sh $rounds $libs 2>&1 | tee $out.txt
    for t in "$@"; do
      if [ "$t" = "-" ]; then unset CTU_ENGINE_LIB; else export CTU_ENGINE_LIB=$V/lib_$t.so; fi
      echo "== parity $t" | tee -a $out.txt
      timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "c1_ or c2_ or fixture or smoke or baseline_sizes" 2>&1 | tail -3 | tee -a $out.txt
    done ;;
  abonly)   # abonly <out> <rounds> <lib tags...>: timing only (diagnostic ablation builds give wrong rows by design)
    out=gpurun_out/$2; rounds=$3; shift 3
    libs=""; for t in "$@"; do if [ "$t" = "-" ]; then libs="$libs -"; else libs="$libs $V/lib_$t.so"; fi; done
    for r in $(seq $rounds); do for L in $libs; do
      if [ "$L" = "-" ]; then unset CTU_ENGINE_LIB; else export CTU_ENGINE_LIB=$L; fi
      echo -n "$L  "; python tools/bench_cfg.py --cfg ${CFG:-C2} --utts ${UTTS:-10000} --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms %.4f kernel %.4f' % (d['ms_per_step'], d['front_kernel_ms']))"
    done; done 2>&1 | tee $out.txt ;;
  tests)
    timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; echo "pytest rc $?"; tail -8 gpurun_out/gpu_tests.log ;;
  *) echo "usage: $0 ab|tests ..."; exit 2 ;;
esac
