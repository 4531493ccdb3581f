# per-kernel durations of one bench.py command (run ON the GPU box):  bash tools/kstats.sh <out.csv> <bench args ...>
OUT=$(realpath -m "$1"); shift
R=$(cd "$(dirname "$0")/.." && pwd)
export TMPDIR=/tmp
rm -rf /tmp/kstats_prof
python $R/bench.py --cpu-frames 0 --no-verify --no-extras --steps 1 --warmup 0 "$@" > /dev/null 2>&1   # (the synthetic stream is rendered by a process pool on first use: not under the profiler)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kstats_prof -- python $R/bench.py --cpu-frames 0 --no-verify "$@" > /dev/null 2>&1 )
cp $(find /tmp/kstats_prof -name '*kernel_stats.csv' | head -1) $OUT
python - "$OUT" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0].replace('orbfe::', '').replace('void ', '')
    print("%-30s calls %4s avg %9.1f us min %9.1f max %9.1f" % (n[:30], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
