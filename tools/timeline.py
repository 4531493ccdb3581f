"""Timeline of one steady-state step from a rocprofv3 --kernel-trace run of bench.py (run ON the GPU box):
    python tools/timeline.py
Prints every kernel of one step in the middle of the run with its queue, start and end (us, relative), so that the
overlap between the engines' streams can be read off."""
import csv, glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, "gpurun_out", "tl")
subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                os.path.join(root, "bench.py"), "--cpu-frames", "0", "--no-verify", "--steps", "8", "--warmup", "2", *sys.argv[1:]],
               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True)
rows = []
for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("orbfe::", "").replace("void ", "")))
rows.sort()
# steps are delimited by k_adaptive_threshold launches; take the 6th
thr = [i for i, r in enumerate(rows) if r[3].startswith("k_adaptive_threshold")]
if len(thr) < 8:
    thr = [i for i, r in enumerate(rows) if r[3].startswith("k_fast_cells")]
a, b = thr[5], thr[6]
t0 = rows[a][0]
qs = sorted({r[2] for r in rows[a:b + 40]})
print("queues:", qs)
for s, e, q, n in rows[a - 2:b + 25]:
    print("%9.1f %9.1f  q%-2d %7.1f us  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, qs.index(q) if q in qs else -1, (e - s) / 1e3, n[:40]))
