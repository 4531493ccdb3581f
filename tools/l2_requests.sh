cd /tmp; export TMPDIR=/tmp
for grp in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum" "TCP_TCC_WRITE_REQ_sum TCC_EA_RDREQ_sum"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/l2req_$(echo $grp | cut -d' ' -f1)
  rm -rf $d
  timeout -k 5 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-verify --no-extras > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('gpurun_out/l2req_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name'].split('(')[0].replace('orbfe::', '').replace('void ', '')[:32]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
names = sorted({c for v in agg.values() for c in v})
print("%-34s %5s " % ("kernel", "n") + " ".join("%14s" % c[:14] for c in names))
tot = collections.Counter()
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('TCP_TCC_READ_REQ_sum', [0])) / max(1, len(kv[1].get('TCP_TCC_READ_REQ_sum', [0])))):
    n = max(len(x) for x in v.values())
    print("%-34s %5d " % (k, n) + " ".join("%14.2f" % (sum(v.get(c, [0])) / max(1, len(v.get(c, [0]))) / 1e6) for c in names))
PY
rm -rf gpurun_out/l2req_*
