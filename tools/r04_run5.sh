# ON THE GPU BOX: C5 and single-frame latency over tile width / tiles per wave of the tiled contour path
mkdir -p gpurun_out
for wt in "320 1" "320 2" "480 1" "480 2" "192 2" "192 4"; do set -- $wt
  ORBFE_ARUCO_TPW=$2 ORBFE_ARUCO_TILE_W=$1 python bench.py --config C5 --cpu-frames 0 --no-verify --steps 10 --out gpurun_out/tw.json > /dev/null 2>&1
  python - <<PY
import json
d=json.load(open("gpurun_out/tw.json")); print("C5 pipeline tile_w $1 tpw $2 ms", round(d["ms_per_step"],3), {k: round(v) for k,v in d["stage_us"].items() if k.startswith("aruco")}, "alone", round(d["roofline"]["stages"]["aruco_contours"].get("launch_us_alone") or 0))
PY
done
for wt in "64 1" "128 1" "192 1" "320 1"; do set -- $wt
  ORBFE_ARUCO_TPW=$2 ORBFE_ARUCO_TILE_W=$1 python bench.py --latency --cpu-frames 0 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('latency tile_w $1', round(d['value'],3), {k: round(v,3) for k,v in d['median_ms'].items()}, 'paired', round(d['paired']['value'],3), {k: round(v,3) for k,v in d['paired']['median_ms'].items()})"
done
ORBFE_ARUCO_TILED=0 python bench.py --latency --cpu-frames 0 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('latency legacy', round(d['value'],3), {k: round(v,3) for k,v in d['median_ms'].items()}, 'paired', round(d['paired']['value'],3), {k: round(v,3) for k,v in d['paired']['median_ms'].items()})"
