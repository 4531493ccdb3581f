mkdir -p gpurun_out
for sp in 1 2; do
ORBFE_H2D_SPLIT=$sp python bench.py --from-host --config C2 --steps 20 --out gpurun_out/fh.json > gpurun_out/fh.log 2>&1 || tail -5 gpurun_out/fh.log
python - <<PY
import json
d=json.load(open("gpurun_out/fh.json")); print("C2 from host split $sp", round(d["value"]), round(d["ms_per_step"],3), round(d["pcie"]["h2d_GBps"],1), round(d["pcie"]["d2h_GBps"],1), d["verified_frames"] and d["verified_frames"]["frames"])
PY
done
