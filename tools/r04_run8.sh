mkdir -p gpurun_out
( time python bench.py --out gpurun_out/r04x_bench.json > gpurun_out/r04x_bench.log 2>&1 ) 2>&1 | grep real; tail -3 gpurun_out/r04x_bench.log | cut -c1-300
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04x_bench.json"))
print("C2", d["value"], d["ms_per_step"], d["verified_frames"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("all_cores"))
e=d.get("extras",{})
print("latency", {k: (round(v,3) if isinstance(v,float) else v) for k,v in e.get("latency",{}).items() if k in ("value","median_ms","error")}, "paired", e.get("latency",{}).get("paired"))
for c in ("C3","C5"): print(c, {k: v for k,v in e.get(c,{}).items() if k not in ("verified_frames",)}, (e.get(c,{}).get("verified_frames") or {}).get("frames"))
PY
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
