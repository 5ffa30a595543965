#!/bin/bash
# usage: tools/ab.sh libA.so libB.so ...   -- interleaved A/B timing of engine builds in one GPU session
for round in 1 2 3; do
for lib in "$@"; do
  CTU_ENGINE_LIB=$PWD/ctucopy_amd/$lib timeout -k 10 300 python bench.py --steps 10 --warmup 2 --utts 2000 --no-cpu 2>/dev/null > /tmp/ab.json
  python - "$lib" <<'PY'
import json,sys
d=json.loads(open("/tmp/ab.json").read().strip().split("\n")[-1])
print(sys.argv[1], "frames/s %.4g" % d["value"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"])
PY
done; done
