#!/bin/bash
# ON THE GPU BOX:  bash tools/ab_many.sh <rounds> "<bench args>" name1=lib1.so name2=lib2.so ...   interleaved, mean and spread of the step
cd "$(dirname "$0")/.."
R=$1; ARGS=$2; shift 2
for r in $(seq $R); do
  for nv in "$@"; do
    name=${nv%%=*}; lib=${nv#*=}
    ms=$(ORBFE_LIB=$PWD/$lib python bench.py --cpu-frames 0 --no-verify --steps 40 $ARGS 2>/dev/null | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$name $ms"
  done
done | sort | awk '{a[$1]=a[$1]" "$2; s[$1]+=$2; q[$1]+=$2*$2; n[$1]++} END {for (k in a) printf "%-12s mean %.4f  sd %.4f  n %d :%s\n", k, s[k]/n[k], sqrt(q[k]/n[k]-(s[k]/n[k])^2), n[k], a[k]}'
