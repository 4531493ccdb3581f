#!/bin/bash
# ON THE GPU BOX: the detector's kernels alone (bench.py --no-orb under rocprofv3) for the four builds / settings of tools/r05_specks.sh
cd "$(dirname "$0")/.."
ARGS=${1:-}
for spec in "new=" "nospecks=ORBFE_ARUCO_SPECKS=0" "nofilter=ORBFE_LIB=$PWD/build/liborbfe_nofilter.so" "old=ORBFE_LIB=$PWD/build/liborbfe_nofilter.so ORBFE_ARUCO_SPECKS=0"; do
  name=${spec%%=*}; envs=${spec#*=}
  echo "== $name"
  env $envs bash tools/kstats.sh gpurun_out/r05a/kalone_$name.csv --no-orb --no-extras $ARGS | grep -i "contours\|speck\|tail\|thresh\|ct_" 
done
