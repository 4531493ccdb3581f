# ON THE GPU BOX: kernel + copy timeline of the single-frame drop-in calls (bench.py --latency): two windows of consecutive dispatches,
# one a quarter into the trace (the plain sequence: extract, detect + poses, search_for_initialization one after the other) and one
# three quarters in (the paired sequence: extract and detect overlapped).  bash tools/latency_profile.sh > gpurun_out/lat_timeline.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lat -- python $GRAFT_REPO_ROOT/bench.py --latency --cpu-frames 0 > $GRAFT_REPO_ROOT/gpurun_out/latency_prof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
cut -c1-900 gpurun_out/latency_prof.json
python - <<'PY'
import csv, glob
rows = []
for fn in glob.glob('gpurun_out/prof_lat/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('orbfe::', '').replace('void ', '')[:40], 'q' + r.get('Queue_Id', '?')))
for fn in glob.glob('gpurun_out/prof_lat/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', r.get('Name', '')), ''))
rows.sort()
for frac in (0.25, 0.75):
    a = int(len(rows) * frac)
    # begin at a host-to-device copy (the start of a call)
    while a < len(rows) and not rows[a][2].startswith('COPY'): a += 1
    t0 = rows[a][0]
    print("---- window at %.0f %% of the trace" % (100 * frac))
    prev_end = t0
    for s, e, n, q in rows[a:a + 110]:
        print("%9.1f %9.1f %7.1f us  gap %6.1f  %-4s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, q, n))
        prev_end = max(prev_end, e)
PY
rm -rf gpurun_out/prof_lat
