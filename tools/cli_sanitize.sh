#!/bin/bash
# Host code of bin/ctucopy under ThreadSanitizer and AddressSanitizer + UBSan on the GPU box (the engine library is not instrumented):
#   g++ -O1 -g -fsanitize=thread|address,undefined ... ctucopy_amd/host/main.cc -o bin/ctucopy_thre|ctucopy_addr
# 60 short files in 1 MiB batches (five hand-overs), then a list with a missing file.
D=/tmp/ctu_san; rm -rf $D; mkdir -p $D/i $D/o
python3 - <<'PY'
import numpy as np
rng = np.random.default_rng(7)
with open("/tmp/ctu_san/list.scp", "w") as f:
    for k in range(60):
        (rng.standard_normal(int(rng.integers(16000, 64000))) * 2000).astype("<i2").tofile("/tmp/ctu_san/i/f%03d.raw" % k)
        f.write("/tmp/ctu_san/i/f%03d.raw /tmp/ctu_san/o/f%03d.htk\n" % (k, k))
PY
for B in bin/ctucopy_thre bin/ctucopy_addr; do
  [ -x $B ] || continue
  echo "== $B"
  export TSAN_OPTIONS="halt_on_error=0 report_signal_unsafe=0" ASAN_OPTIONS="detect_leaks=1:protect_shadow_gap=0"
  setarch x86_64 -R $B -fs 16000 -format_in raw -format_out htk -preset mfcc -S $D/list.scp --batch-mib 1 --io-threads 8 --write-threads 3 > $D/log_$(basename $B) 2>&1; echo "rc $?"
  L=$D/log_$(basename $B)
  echo "reports (every frame of every report counts, not only main.cc's): $(grep -c "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error" $L)"
  grep -A12 "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error" $L | grep "main.cc" | sort | uniq -c | head -20
  ls $D/o | wc -l
  mv $D/i/f030.raw $D/i/x; setarch x86_64 -R $B -fs 16000 -format_in raw -format_out htk -preset mfcc -S $D/list.scp --batch-mib 1 > $D/log2_$(basename $B) 2>&1; echo "rc $? (missing file)"; mv $D/i/x $D/i/f030.raw
  grep -c "WARNING: ThreadSanitizer\|ERROR: AddressSanitizer\|runtime error" $D/log2_$(basename $B); grep -B2 -A14 'Direct leak' $D/log2_$(basename $B) | grep 'main.cc\|Direct leak' | head -12; tail -2 $D/log2_$(basename $B) | cut -c1-200
done
mkdir -p gpurun_out/cli_sanitize_logs && for f in $D/log_* $D/log2_*; do head -c 200000 $f > gpurun_out/cli_sanitize_logs/$(basename $f).txt; done
