# ON THE GPU BOX: contour tests, then ArUco alone / pipeline for tile widths and tiles per wave
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_aruco_gpu.py tests/test_aruco_modes_gpu.py -m gpu -x -q 2>&1 | tail -5
for c in C2 C3; do
for w in 96 128 192 320; do for t in 2 4 8; do
  ORBFE_ARUCO_TPW=$t ORBFE_ARUCO_TILE_W=$w python bench.py --config $c --cpu-frames 0 --no-verify --no-orb --steps 10 --out gpurun_out/tw.json > /dev/null 2>&1
  python - <<PY
import json
d=json.load(open("gpurun_out/tw.json")); print("$c alone tile_w $w tpw $t ms", round(d["ms_per_step"],3), {k: round(v) for k,v in d["stage_us"].items() if k.startswith("aruco")})
PY
done; done; done
bash tools/quick.sh r04u ""
