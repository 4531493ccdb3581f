mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q 2>&1 | tail -15
python bench.py --cpu-frames 0 --out gpurun_out/r04w_C2.json > gpurun_out/r04w_C2.log 2>&1 || tail -20 gpurun_out/r04w_C2.log
python bench.py --gpus 1 --force-gather --cpu-frames 0 --out gpurun_out/r04w_C2_gather.json > gpurun_out/r04w_C2g.log 2>&1 || tail -20 gpurun_out/r04w_C2g.log
python - <<'PY'
import json
for c in ("C2","C2_gather"):
    d=json.load(open("gpurun_out/r04w_%s.json" % c)); print(c, d.get("value"), d["ms_per_step"], d["verified_frames"], d["config"]["env_nondefault"], d.get("gather_check"), d.get("gather_us"), "host_enqueue_ms", d["host_enqueue_ms_per_step"])
PY
