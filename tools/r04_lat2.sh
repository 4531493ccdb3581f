# ON THE GPU BOX: kernels of 40 single-frame detector calls (time per kernel and the gaps between them)
mkdir -p gpurun_out; export TMPDIR=/tmp; R=$PWD
rm -rf /tmp/latk; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/latk -- python $R/tools/r04_lat_kernels.py > /dev/null 2>&1 )
T=$(find /tmp/latk -name '*kernel_trace.csv' | head -1)
python - "$T" <<'PY'
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0].replace('orbfe::','').replace('void ','') for r in rows]
cuts = [i for i,n in enumerate(names) if n.startswith('k_adaptive_threshold')]
agg = collections.OrderedDict(); span = 0.0; busy = 0.0; calls = 0
for c0, c1 in list(zip(cuts, cuts[1:]))[-20:]:
    for k in range(c0, c1):
        d = (int(rows[k]['End_Timestamp'])-int(rows[k]['Start_Timestamp']))/1e3
        a = agg.setdefault(names[k], [0.0, 0]); a[0] += d; a[1] += 1; busy += d
    span += (int(rows[c1]['Start_Timestamp'])-int(rows[c0]['Start_Timestamp']))/1e3
    calls += 1
for n,(d,c) in agg.items():
    print("%-34s x%4.1f %7.1f us per call" % (n[:34], c/calls, d/calls))
print("kernel time per call %.1f us, call period %.1f us" % (busy/calls, span/calls))
PY
