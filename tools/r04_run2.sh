mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
bash tools/quick.sh r04tiled ""
ORBFE_ARUCO_TILED=0 python bench.py --cpu-frames 0 --no-verify --out gpurun_out/r04_legacy.json > /dev/null 2>&1; python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_legacy.json")); print("legacy relay:", d["diagnostic_frames_per_s"], d["ms_per_step"], {k: round(v) for k,v in d["stage_us"].items() if k.startswith("aruco")})
PY
for c in C3 C5; do python bench.py --config $c --cpu-frames 0 --steps 10 --out gpurun_out/r04tiled_$c.json > /dev/null 2>&1; ORBFE_ARUCO_TILED=0 python bench.py --config $c --cpu-frames 0 --steps 10 --no-verify --out gpurun_out/r04legacy_$c.json > /dev/null 2>&1; python - <<PY
import json
for n in ("tiled","legacy"):
    d=json.load(open("gpurun_out/r04%s_$c.json" % n)); print("$c", n, d.get("value") or d.get("diagnostic_frames_per_s"), d["ms_per_step"], d["verified_frames"] and d["verified_frames"]["frames"], {k: round(v) for k,v in d["stage_us"].items() if k.startswith("aruco")}, {k: round(d["roofline"]["stages"][k].get("launch_us_alone") or 0) for k in d["roofline"]["stages"] if k.startswith("aruco")})
PY
done
python bench.py --latency --cpu-frames 0 --out gpurun_out/r04tiled_latency.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('latency', d['value'], d['median_ms'], 'paired', d['paired']['value'], d['paired']['median_ms'])"
