# ON THE GPU BOX: the contour tests, then C2 / C3 / C5 bench lines (tiled default) with kernel stats of C2
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_aruco_gpu.py tests/test_aruco_modes_gpu.py tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -5
bash tools/quick.sh r04t ""
for c in C3 C5; do python bench.py --config $c --cpu-frames 0 --steps 10 --out gpurun_out/r04t_$c.json > /dev/null 2>&1; python - <<PY
import json
d=json.load(open("gpurun_out/r04t_$c.json")); print("$c", d.get("value") or d.get("diagnostic_frames_per_s"), d["ms_per_step"], d["verified_frames"] and d["verified_frames"]["frames"], {k: round(v) for k,v in d["stage_us"].items() if k.startswith("aruco")}, {k: round(d["roofline"]["stages"][k].get("launch_us_alone") or 0) for k in d["roofline"]["stages"] if k.startswith("aruco")})
PY
done
python bench.py --latency --cpu-frames 0 --out gpurun_out/r04t_latency.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('latency', d['value'], d['median_ms'], 'paired', d['paired']['value'], d['paired']['median_ms'])"
