"""Kernel trace of a short bench run as a compact table (run ON the GPU box); analysed offline with tools/trace_view.py:
    python tools/trace_dump.py <out.tsv> [bench args]
Columns: start_us end_us queue kernel (times relative to the first kernel)."""
import csv, glob, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = sys.argv[1]
d = "/tmp/trace_dump_%d" % os.getpid()
subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                os.path.join(root, "bench.py"), "--cpu-frames", "0", "--no-verify", "--no-extras", "--steps", "24", "--warmup", "5", *sys.argv[2:]],
               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True)
rows = []
for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"].split("(")[0].replace("orbfe::", "").replace("void ", "")))
rows.sort()
t0 = rows[0][0] if rows else 0
qs = sorted({r[2] for r in rows})
with open(out, "w") as fh:
    for s, e, q, n in rows:
        fh.write("%.1f\t%.1f\t%d\t%s\n" % ((s - t0) / 1e3, (e - t0) / 1e3, qs.index(q), n[:48]))
shutil.rmtree(d, ignore_errors=True)
print(len(rows), "kernels ->", out)
