#!/bin/bash
# ON THE GPU BOX: timeline of bench.py --from-host (kernels and memory copies) -- which of upload, engines and read-back overlap
cd "$(dirname "$0")/.."
R=$PWD; export TMPDIR=/tmp
rm -rf /tmp/fh; mkdir -p /tmp/fh
python bench.py --from-host --cpu-frames 0 --no-verify --steps 2 --warmup 1 > /dev/null 2>&1
( cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/fh -- python $R/bench.py --from-host --cpu-frames 0 --no-verify --steps 6 --warmup 1 > /tmp/fh/out.txt 2> /tmp/fh/err.txt ); echo "rc=$?"
tail -2 /tmp/fh/out.txt | cut -c1-300; tail -3 /tmp/fh/err.txt | cut -c1-200
K=$(find /tmp/fh -name '*kernel_trace.csv' | head -1); M=$(find /tmp/fh -name '*memory_copy_trace.csv' | head -1)
head -2 "$M"
python - "$K" "$M" <<'PY'
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K q%s %s' % (r.get('Queue_Id', '?'), r['Kernel_Name'].split('(')[0].replace('orbfe::', '').replace('void ', '')[:28])))
for r in csv.DictReader(open(sys.argv[2])):
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY %s' % r.get('Direction', '?')))
ev.sort()
# the last steps: from the fourth-last long host-to-device copy on
big = [i for i, e in enumerate(ev) if e[2].startswith('COPY') and 'HOST_TO_DEVICE' in e[2].upper() and e[1] - e[0] > 500_000]
i0 = big[-4] if len(big) >= 4 else 0
t0 = ev[i0][0]
for s, e, n in ev[i0:]:
    if (n.startswith('COPY') and e - s > 20_000) or (e - s) > 60_000:
        print('%9.1f %9.1f  %8.1f us  %s' % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
