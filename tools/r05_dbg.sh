#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
rm -rf /tmp/kp; mkdir -p /tmp/kp
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kp -- python $R/bench.py --cpu-frames 0 --no-verify --no-extras --steps 6 --warmup 2 > /tmp/kp/out.txt 2> /tmp/kp/err.txt ); echo "rc=$?"
tail -5 /tmp/kp/out.txt | cut -c1-300; tail -20 /tmp/kp/err.txt | cut -c1-300
find /tmp/kp -name '*.csv' | head
f=$(find /tmp/kp -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f gpurun_out/r05a/kfull_new.csv && python tools/kstat_summary.py $f 30
