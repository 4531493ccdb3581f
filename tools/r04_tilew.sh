# ON THE GPU BOX: tile width of k_ct_walk, ArUco alone and in the pipeline
mkdir -p gpurun_out
for c in C2 C3; do
for w in 128 192 320 640 960; do
  ORBFE_ARUCO_TILE_W=$w python bench.py --config $c --cpu-frames 0 --no-verify --no-orb --steps 10 --out gpurun_out/tw.json > /dev/null 2>&1
  python - <<PY
import json
d=json.load(open("gpurun_out/tw.json")); print("$c alone tile_w $w ms", round(d["ms_per_step"],3), {k: round(v) for k,v in d["stage_us"].items() if k.startswith("aruco")})
PY
  ORBFE_ARUCO_TILE_W=$w python bench.py --config $c --cpu-frames 0 --no-verify --steps 10 --out gpurun_out/tw.json > /dev/null 2>&1
  python - <<PY
import json
d=json.load(open("gpurun_out/tw.json")); print("$c pipeline tile_w $w ms", round(d["ms_per_step"],3), {k: round(v) for k,v in d["stage_us"].items() if k.startswith("aruco")})
PY
done
done
