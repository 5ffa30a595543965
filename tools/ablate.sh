#!/bin/bash
# usage: tools/ablate.sh "<modes>"  -- times bench.py under CTU_DEBUG_MODE values (diagnostic kernel ablations)
for m in ${1:-0 1 2}; do
  CTU_DEBUG_MODE=$m timeout -k 10 300 python bench.py --steps 5 --warmup 2 --utts 2000 --no-cpu 2>/dev/null > /tmp/ab_$m.json
  python - "$m" <<'PY'
import json,sys
m=sys.argv[1]
d=json.loads(open(f"/tmp/ab_{m}.json").read().strip().split("\n")[-1])
print("mode", m, "frames/s %.4g" % d["value"], "kernel_ms %.4f" % d["roofline"]["kernel_ms"], "frac %.4f" % d["roofline"]["frac"])
PY
done
