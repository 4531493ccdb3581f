# ON THE GPU BOX: which resource is the pipeline's step time made of?  tools/ballast.hip next to every step, one resource at a time, at
# sizes that keep the resource busy for roughly 100 / 200 us alone; prints ms per step and the ballast's own duration alone / next to a step.
#     bash tools/ballast_sweep.sh [rounds] ["<bench args>"]
cd "$(dirname "$0")/.."
[ -f build/libballast.so ] || hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o build/libballast.so tools/ballast.hip
R=${1:-2}; ARGS=$2
for r in $(seq $R); do
  for spec in none valu:${V1:-1000} valu:${V2:-2000} ta:${T1:-1000} ta:${T2:-2000} l2:${L1:-300} l2:${L2:-600} lds:${D1:-1000} lds:${D2:-2000} hbm:${H1:-800} hbm:${H2:-1600}; do
    if [ $spec = none ]; then e=""; else e="ORBFE_BENCH_BALLAST=$spec"; fi
    env $e timeout -k 5 300 python bench.py --cpu-frames 0 --no-verify --no-extras $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); b = d.get('ballast') or {}
print('%-12s %.4f ms/step   ballast alone %7.1f us, next to a step %7.1f us' % ('$spec', d['ms_per_step'], b.get('us_alone', 0), b.get('us_next_to_a_step', 0)))"
  done
done
