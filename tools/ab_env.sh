#!/bin/bash
# ON THE GPU BOX:  bash tools/ab_env.sh <rounds> "<bench args>" "NAME=ENV1=v ENV2=v" ...   interleaved A/B of environment settings
# prints ms_per_step of every run (value is null for non-default settings; the step time is what is compared)
cd "$(dirname "$0")/.."
R=$1; ARGS=$2; shift 2
for r in $(seq $R); do
  for spec in "base=" "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    ms=$(env $envs python bench.py --cpu-frames 0 --no-verify $ARGS 2>/dev/null | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$name $ms"
  done
done | sort | awk '{a[$1]=a[$1]" "$2; s[$1]+=$2; n[$1]++} END {for (k in a) printf "%-24s mean %.4f  runs%s\n", k, s[k]/n[k], a[k]}'
