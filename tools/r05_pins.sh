#!/bin/bash
# ON THE GPU BOX: lock stages again with the round-5 detector (bash tools/r05_pins.sh [rounds])
cd "$(dirname "$0")/.."
R=${1:-2}
run() { name=$1; shift; env "$@" timeout -k 5 200 python bench.py --cpu-frames 0 --no-verify --no-extras $EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); s = d.get('stage_us') or {}
print('$name %.4f' % d['ms_per_step'], {k: round(v) for k, v in s.items()})"; }
EXTRA="--no-aruco" run noaruco A=1
for r in $(seq $R); do
  run new A=1
  run old ORBFE_LIB=$PWD/build/liborbfe_nofilter.so ORBFE_ARUCO_SPECKS=0
  for dp in 0 1 2 3 12 13; do run new_detpin$dp ORBFE_DET_PIN=$dp; done
  for pp in 0 1 3; do run new_phasepin$pp ORBFE_PHASE_PIN=$pp; done
  run new_nodefer ORBFE_DEFER_POST=0
  run new_fork ORBFE_DET_NOFORK=0
done
