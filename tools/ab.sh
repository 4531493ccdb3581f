#!/bin/bash
# ON THE GPU BOX: interleaved A/B of anything that is an environment setting -- a library build (ORBFE_LIB=build/liborbfe_x.so, made with
# tools/build_variant.sh), an ORBFE_* switch, or both:
#     bash tools/ab.sh <rounds> "<bench args>" "name=ENV=v ENV2=v ..." ["name2=..."] ...
# Every round runs every variant once ("base=" = no setting is always the first), so box-to-box and minute-to-minute drift hits all
# of them alike.  Prints mean / sd / min / max of ms_per_step and the runs; STAGES=1 adds the last run's stage times.
# Examples (the sweeps quoted in DESIGN.md are all of this form; tools/sweeps.md lists them):
#     bash tools/ab.sh 4 "" "pin1=ORBFE_PHASE_PIN=1" "pin3=ORBFE_PHASE_PIN=3"
#     bash tools/ab.sh 3 "--config C3" "rows4=ORBFE_ARUCO_BAND_ROWS=4" "new=ORBFE_LIB=$PWD/build/liborbfe_new.so"
cd "$(dirname "$0")/.."
R=$1; ARGS=$2; shift 2
for r in $(seq $R); do
  for spec in "base=" "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    env $envs timeout -k 5 300 python bench.py --cpu-frames 0 --no-verify --no-extras $ARGS 2>/dev/null | STAGES=${STAGES:-0} python -c "
import sys, json, os
try:
    d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]
    print('$name %.4f' % d['ms_per_step'], json.dumps({k: round(v) for k, v in (d.get('stage_us') or {}).items()}) if os.environ['STAGES'] == '1' else '')
except Exception as e:
    print('$name nan', repr(e))"
  done
done | sort -s -k1,1 | awk '{n[$1]++; v=$2+0; s[$1]+=v; q[$1]+=v*v; a[$1]=a[$1]" "$2; if (!($1 in mn) || v<mn[$1]) mn[$1]=v; if (v>mx[$1]) mx[$1]=v; st[$1]=$0}
  END {for (k in a) {m=s[k]/n[k]; printf "%-22s mean %.4f sd %.4f min %.4f max %.4f n %d :%s\n", k, m, sqrt(q[k]/n[k]-m*m>0?q[k]/n[k]-m*m:0), mn[k], mx[k], n[k], a[k]; if (ENVIRON["STAGES"]=="1") {sub(/^[^ ]+ [^ ]+ /,"",st[k]); print "    " st[k]}}}' | sort
