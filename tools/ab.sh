# A/B on one box: step time and stage times of bench.py for several builds of the library, interleaved, best of 3.
#   bash tools/ab.sh "<bench args>" name1=path1.so name2=path2.so ...
cd "$(dirname "$0")/.."
ARGS="$1"; shift
for rep in 1 2 3; do
for nv in "$@"; do
  name=${nv%%=*}; lib=${nv#*=}
  ORBFE_LIB=$PWD/$lib python bench.py --cpu-frames 0 --no-verify $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-10s %.3f ms' % ('$name', d['ms_per_step']), {k:round(v) for k,v in d['stage_us_last_step'].items()})"
done
done
