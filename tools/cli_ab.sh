#!/bin/bash
# File-to-file A/B of the command-line host on the GPU box: the serial host loop of round 3's first half (bin/ctucopy_r3serial,
# if present) against the pipelined one, then the pipeline's knobs.  usage: bash tools/cli_ab.sh [utts]
N=${1:-10000}
D=/tmp/ctu_e2e
O=gpurun_out/cli_e2e_r03.txt
: > $O
if [ -x bin/ctucopy_r3serial ]; then
  python3 tools/cli_e2e.py --utts $N --runs 2 --bin bin/ctucopy_r3serial --dir $D >> $O 2>&1
  rm -rf $D/out_serial && cp -r $D/out $D/out_serial
fi
python3 tools/cli_e2e.py --utts $N --runs 3 --bin bin/ctucopy --dir $D >> $O 2>&1
if [ -d $D/out_serial ]; then diff -rq $D/out $D/out_serial > /dev/null && echo "outputs identical to the serial host's" >> $O || echo "OUTPUTS DIFFER" >> $O; fi
for X in "--write-threads 2" "--write-threads 4" "--io-threads 4" "--io-threads 1" "--batch-mib 64" "--batch-mib 128" "--batch-mib 1024"; do
  python3 tools/cli_e2e.py --utts $N --runs 2 --bin bin/ctucopy --dir $D -- $X 2>/dev/null >> $O
done
echo "nproc $(nproc), cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" >> $O
cat $O
