#!/bin/bash
cd $GRAFT_REPO_ROOT
for d2 in 0 1 2; do
echo "phase 2 only, MM, A operand variant $d2"
CTU_DEBUG2=$d2 CTU_DEBUG_MODE=2 python tools/bench_cfg.py --cfg C2 --steps 10 | cut -c1-120
echo "full kernel"
CTU_DEBUG2=$d2 python tools/bench_cfg.py --cfg C2 --steps 10 | cut -c1-120
done
