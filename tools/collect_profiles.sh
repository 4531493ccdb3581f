#!/bin/bash
# HERE (the build container), after `tools/gpu.sh -- 'bash tools/profile_round.sh <tag>'` has merged gpurun_out/<tag>/ back:
#     bash tools/collect_profiles.sh <tag>
# copies what is to be judged into profiles/ -- the round's tagged files AND the per-configuration profiles bench.py reads
# (traffic_C*.json / pmc_stage_C*.json, written on the box by tools/make_profiles.py), all three configurations, so that no
# configuration keeps an earlier round's counters next to this round's.
TAG=${1:?tag}
cd "$(dirname "$0")/.."
O=gpurun_out/$TAG
[ -d "$O" ] || { echo "no $O"; exit 1; }
cp $O/${TAG}_* profiles/
for c in C2 C3 C5; do
  for f in traffic_$c.json pmc_stage_$c.json; do
    if [ -f "$O/$f" ]; then cp "$O/$f" profiles/$f; else echo "MISSING $O/$f: removing the stale profiles/$f"; rm -f profiles/$f; fi
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("profiles/traffic_C*.json") + glob.glob("profiles/pmc_stage_C*.json")):
    d = json.load(open(f)); print(f, d.get("_profile"), d.get("_commit"), d.get("_library_sha16"))
PY
