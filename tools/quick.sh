#!/bin/bash
# ON THE GPU BOX:  bash tools/quick.sh <tag> "<pytest args or empty>" [bench args]
# a test subset, one default bench line and the rocprofv3 kernel-trace summary of the same command, under gpurun_out/<tag>_*
TAG=$1; TESTS=$2; shift 2
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
if [ -n "$TESTS" ]; then python -m pytest $TESTS -m gpu -x -q 2>&1 | tail -12; fi
python bench.py --cpu-frames 0 "$@" --out gpurun_out/${TAG}_bench.json > gpurun_out/${TAG}_bench.log 2>&1 || tail -20 gpurun_out/${TAG}_bench.log
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print("value", d["value"], "ms_per_step", d["ms_per_step"], "verified", d["verified_frames"] and d["verified_frames"]["frames"])
print({k: round(v) for k, v in d["stage_us"].items()})
PY
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python $R/bench.py --cpu-frames 0 --no-verify "$@" > /dev/null 2>&1 )
cp $(find gpurun_out/prof_$TAG -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG
python tools/kstat_summary.py gpurun_out/${TAG}_kernel_stats.csv 24
