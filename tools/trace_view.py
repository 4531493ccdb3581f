"""Offline view of tools/trace_dump.py output: python tools/trace_view.py <trace.tsv> [first_step [nsteps]]
Prints the kernels of a few steady-state steps per hardware queue side by side in time order, and the step period."""
import sys
rows = [l.rstrip("\n").split("\t") for l in open(sys.argv[1])]
rows = [(float(a), float(b), int(q), n) for a, b, q, n in rows]
fast = [r for r in rows if r[3].startswith("k_fast_cells")]
per = [fast[i + 1][0] - fast[i][0] for i in range(len(fast) - 1)]
print("k_fast_cells starts:", " ".join("%.0f" % f[0] for f in fast))
print("periods:", " ".join("%.0f" % p for p in per))
k0 = int(sys.argv[2]) if len(sys.argv) > 2 else 6
ns = int(sys.argv[3]) if len(sys.argv) > 3 else 2
t0, t1 = fast[k0][0] - 700, fast[k0 + ns][0] + 100
for a, b, q, n in rows:
    if a < t0 or a > t1: continue
    print("%9.1f %9.1f %s q%d %7.1f  %s" % (a - t0, b - t0, "          " * q, q, b - a, n))
