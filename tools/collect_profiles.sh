#!/bin/bash
# HERE (build container), after `tools/gpu.sh -- 'bash tools/profile_round.sh <tag>'`: copy the round's files from gpurun_out/<tag>/ into
# profiles/ and regenerate the per-configuration counter profiles with the current stage map, keeping the ids stamped on the GPU box.
TAG=${1:-r03}
cd "$(dirname "$0")/.."
for f in ${TAG}_bench.json ${TAG}_bench_c3.json ${TAG}_bench_c5.json ${TAG}_latency.json ${TAG}_bench_2ranks_gloo_one_gpu.json ${TAG}_bench_rccl_world1_force_gather.json ${TAG}_bench_c4_8ranks_gloo_one_gpu.json ${TAG}_kernel_stats.csv ${TAG}_pmc_table.txt ${TAG}_pmc_table_c3.txt ${TAG}_pmc_summary.json ${TAG}_pmc_summary_c3.json ${TAG}_pmc_calibration.json ${TAG}_timeline_full.txt ${TAG}_pytest_gpu.log; do cp gpurun_out/$TAG/$f profiles/ 2>/dev/null || echo missing $f; done
python tools/make_profiles.py gpurun_out/$TAG/${TAG}_pmc_summary.json gpurun_out/$TAG/${TAG}_pmc_calibration.json C2 $TAG > /dev/null
python tools/make_profiles.py gpurun_out/$TAG/${TAG}_pmc_summary_c3.json gpurun_out/$TAG/${TAG}_pmc_calibration.json C3 ${TAG}_c3 > /dev/null
python - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
for c in ("C2", "C3"):
    for nm in ("pmc_stage", "traffic"):
        old = json.load(open("gpurun_out/%s/%s_%s.json" % (tag, nm, c)))
        new = json.load(open("profiles/%s_%s.json" % (nm, c)))
        for k in ("_commit", "_library_sha16", "_profile"):
            new[k] = old.get(k)
        json.dump(new, open("profiles/%s_%s.json" % (nm, c), "w"), indent=1)
for n in ("bench.json", "bench_c3.json", "bench_c5.json", "bench_rccl_world1_force_gather.json"):
    d = json.load(open("profiles/%s_%s" % (tag, n))); r = d["roofline"]
    print(n, round(d["value"]), round(d["ms_per_step"], 3), r["kernel"], round(r["frac"], 4), "fused", round(r["step"]["fused"]["frac"], 4),
          "traffic x", r["step"].get("counter_traffic", {}).get("over_fused"), d["config"]["library_sha16"])
l = json.load(open("profiles/%s_latency.json" % tag)); print("latency", l["value"], l["median_ms"])
PY
