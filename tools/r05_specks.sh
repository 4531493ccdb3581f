#!/bin/bash
# ON THE GPU BOX: the contour stage with / without the speck passes and the run tests on the start candidates (round 5):
#   bash tools/r05_specks.sh [rounds] [bench args]     (needs build/liborbfe_nofilter.so = tools/build_variant.sh nofilter -DORBFE_CAND_FILTER=0)
cd "$(dirname "$0")/.."
R=${1:-3}; ARGS=${2:-}
for r in $(seq $R); do
  for spec in "new=" "specks_unused=ORBFE_ARUCO_SPECKS=2" "nospecks=ORBFE_ARUCO_SPECKS=0" "nofilter=ORBFE_LIB=$PWD/build/liborbfe_nofilter.so" "old=ORBFE_LIB=$PWD/build/liborbfe_nofilter.so ORBFE_ARUCO_SPECKS=0"; do
    name=${spec%%=*}; envs=${spec#*=}
    env $envs timeout -k 5 200 python bench.py --cpu-frames 0 --no-verify --no-extras $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d.get('stage_us') or {}
print('$name %.4f' % d['ms_per_step'], {k: round(v) for k, v in s.items() if k.startswith('aruco')})"
  done
done
