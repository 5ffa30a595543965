#!/bin/bash
# usage: tools/ab_bench.sh <rounds> <lib1> <lib2> ...   -- interleaved A/B of engine builds on the headline workload (bench.py, 10 000 utterances; "-" = the in-tree library)
ROUNDS=$1; shift
for r in $(seq $ROUNDS); do
  for L in "$@"; do
    if [ "$L" = "-" ]; then unset CTU_ENGINE_LIB; else export CTU_ENGINE_LIB=$L; fi
    echo -n "$L  "; python bench.py --no-cpu --no-extra --steps 10 --warmup 2 | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ms %.4f kernel %.4f value %.4g frac %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['roofline']['frac']))"
  done
done
