#!/bin/bash
# usage: tools/ab.sh <cfg> <rounds> <lib1> <lib2> ...   -- interleaved A/B of engine builds on one device ("-" = the in-tree library)
CFG=$1; ROUNDS=$2; shift 2
for r in $(seq $ROUNDS); do
  for L in "$@"; do
    if [ "$L" = "-" ]; then unset CTU_ENGINE_LIB; else export CTU_ENGINE_LIB=$L; fi
    echo -n "$L  "; python tools/bench_cfg.py --cfg $CFG --steps 10 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms %.4f kernel %.4f' % (d['ms_per_step'], d['front_kernel_ms']))"
  done
done
