cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lat -- python $GRAFT_REPO_ROOT/bench.py --latency --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/latency_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
cat gpurun_out/latency_prof.json | cut -c1-600
for f in $(find gpurun_out/prof_lat -name '*stats.csv'); do echo $f; head -40 $f | cut -c1-140; done
python - <<'PY'
import csv, glob
fn = glob.glob('gpurun_out/prof_lat/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# one steady-state frame: find the last k_resize_tab sequence start
names = [r['Kernel_Name'].split('(')[0].replace('orbfe::','').replace('void ','') for r in rows]
idx = [i for i,n in enumerate(names) if n.startswith('k_resize_tab')]
# start of the last frame = resize index where previous kernel is not resize, take third last group
starts = [i for i in idx if i == 0 or not names[i-1].startswith('k_resize_tab')]
a, b = starts[-3], starts[-2]
t0 = int(rows[a]['Start_Timestamp'])
for r, n in zip(rows[a:b], names[a:b]):
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print("%9.1f %9.1f %7.1f us  %s" % (s/1e3, e/1e3, (e-s)/1e3, n[:40]))
PY
rm -rf gpurun_out/prof_lat
