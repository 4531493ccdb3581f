# A/B on one box, interleaved, N rounds: only the step times.   bash tools/ab2.sh N "<bench args>" name=lib.so ...
cd "$(dirname "$0")/.."
N=$1; ARGS="$2"; shift 2
for rep in $(seq $N); do
for nv in "$@"; do
  name=${nv%%=*}; lib=${nv#*=}
  ORBFE_LIB=$PWD/$lib timeout -k 5 200 python bench.py --cpu-frames 0 --no-verify --no-extras $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %.3f' % ('$name', d['ms_per_step']))"
done
done | sort | awk '{a[$1]=a[$1]" "$2; s[$1]+=$2; n[$1]++} END {for (k in a) printf "%s mean %.3f :%s\n", k, s[k]/n[k], a[k]}'
