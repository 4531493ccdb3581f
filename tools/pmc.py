"""Per-kernel hardware counters of one bench.py step, one rocprofv3 --pmc pass per counter group (run ON the GPU box):
    python tools/pmc.py [bench args ...]      e.g.  python tools/pmc.py --no-aruco
Writes gpurun_out/pmc_summary.json and prints a table: per kernel the per-dispatch mean of every counter plus derived
figures (VALU lane utilisation, issue-busy fraction, wait fraction, HBM bytes)."""
import csv, glob, json, os, subprocess, sys, collections

GROUPS = [
    ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU"],
    ["SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_THREAD_CYCLES_VALU"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_LDS"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_ACTIVE_INST_VMEM"],
    ["FETCH_SIZE"], ["WRITE_SIZE"], ["GRBM_GUI_ACTIVE"],
]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", "pmc")
args = sys.argv[1:]
import shutil
shutil.rmtree(out, ignore_errors=True)      # rocprofv3 adds a directory per run: counters of an earlier configuration must not be averaged in
agg = collections.defaultdict(lambda: collections.defaultdict(list))
env = dict(os.environ, TMPDIR="/tmp")
for gi, grp in enumerate(GROUPS):
    d = os.path.join(out, "g%d" % gi)
    cmd = ["rocprofv3", "--pmc", *grp, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
           sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-frames", "0", "--no-verify", *args]
    r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode:
        print("group", grp, "failed:", r.stderr[-400:])
        continue
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            k = row["Kernel_Name"].split("(")[0].replace("orbfe::", "").replace("void ", "")
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/fetch_calib.hip; MI355X_MICROARCH.md, HBM section)
calib = {}
exe = os.path.join(root, "build", "fetch_calib")
if not os.path.exists(exe):
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(root, "tools", "fetch_calib.hip")], capture_output=True)
if os.path.exists(exe):
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out, "calib_" + ctr)
        r = subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "--", exe], cwd="/tmp", env=env,
                           capture_output=True, text=True)
        if r.returncode:
            print("calibration", ctr, "failed:", r.stderr[-300:])
            continue
        acc = collections.defaultdict(list)
        for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(fn)):
                if "calib_" in row["Kernel_Name"]:
                    acc[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            calib.setdefault(k, {})[ctr] = (sum(v) / len(v)) * 1024.0 / float(1 << 30)   # counted bytes per real byte
    json.dump(calib, open(os.path.join(root, "gpurun_out", "pmc_calibration.json"), "w"), indent=1)
    print("FETCH_SIZE / WRITE_SIZE x 1024 per byte really moved (1 GiB streamed once per kernel):")
    for k, v in sorted(calib.items()):
        print("  %-40s %s" % (k, {c: round(x, 3) for c, x in v.items()}))
summary = {}
for k, cs in agg.items():
    summary[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    summary[k]["dispatches"] = max(len(v) for v in cs.values())
json.dump(summary, open(os.path.join(root, "gpurun_out", "pmc_summary.json"), "w"), indent=1)
print("VALUus = VALU wave-instructions x 4 cycles / (1024 SIMDs x 2.4 GHz): the time the kernel needs if it only issued VALU")
print("%-28s %5s %9s %9s %9s %7s %7s %7s %7s %9s %9s %8s" % ("kernel", "n", "VALU/wv", "SALU/wv", "VMEM/wv", "LDS/wv", "lane%", "busy%", "wait%", "fetchMB", "writeMB", "VALUus"))
for k, s in sorted(summary.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
    wv = max(s.get("SQ_WAVES", 1), 1)
    g = lambda n: s.get(n, 0.0)
    lane = 100 * g("SQ_THREAD_CYCLES_VALU") / max(64 * g("SQ_ACTIVE_INST_VALU"), 1) if g("SQ_ACTIVE_INST_VALU") else 0
    busy = 100 * g("SQ_ACTIVE_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1)
    wait = 100 * g("SQ_WAIT_INST_ANY") / max(g("SQ_WAVE_CYCLES"), 1)
    print("%-28s %5d %9.0f %9.0f %9.1f %7.1f %7.1f %7.1f %7.1f %9.1f %9.1f %8.1f" % (
        k[:28], s["dispatches"] // 3, g("SQ_INSTS_VALU") / wv, g("SQ_INSTS_SALU") / wv, (g("SQ_INSTS_VMEM_RD") + g("SQ_INSTS_VMEM_WR")) / wv,
        g("SQ_INSTS_LDS") / wv, lane, busy, wait, g("FETCH_SIZE") / 1024, g("WRITE_SIZE") / 1024,
        g("SQ_INSTS_VALU") * 4 / 1024 / 2400))
