mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
python bench.py --cpu-frames 0 --out gpurun_out/r04v_C2.json > /dev/null 2>&1
python bench.py --config C5 --cpu-frames 0 --steps 10 --out gpurun_out/r04v_C5.json > /dev/null 2>&1
python - <<'PY'
import json
for c in ("C2","C5"):
    d=json.load(open("gpurun_out/r04v_%s.json" % c)); print(c, d.get("value"), d["ms_per_step"], d["verified_frames"] and d["verified_frames"]["frames"], d["config"]["env_nondefault"], {k: round(v) for k,v in d["stage_us"].items() if k.startswith("aruco")})
PY
