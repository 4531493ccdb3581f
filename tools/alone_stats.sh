cd /tmp; export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/tools/det_alone.py 2 > /dev/null 2>&1   # stream cache
for t in det ext; do
  rm -rf /tmp/prof_$t
  timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$t -- python $GRAFT_REPO_ROOT/tools/${t}_alone.py 8 > /dev/null 2>&1
  f=$(find /tmp/prof_$t -name '*kernel_stats.csv' | head -1)
  echo "== $t alone: per-launch average us (calls)"
  python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:22]:
    print("%-34s %8.1f us  x %s  (min %.1f)" % (r['Name'].split('(')[0].replace('orbfe::','').replace('void ','')[:34], float(r['AverageNs'])/1e3, r['Calls'], float(r['MinNs'])/1e3))
PY
done
