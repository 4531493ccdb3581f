mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "from_host or cpp_without" 2>&1 | tail -8
for c in C2 C3; do
python bench.py --from-host --config $c --steps 20 --out gpurun_out/r04_from_host_$c.json > gpurun_out/fh.log 2>&1 || tail -5 gpurun_out/fh.log
python - <<PY
import json
d=json.load(open("gpurun_out/r04_from_host_$c.json")); print("$c from host", round(d["value"]), round(d["ms_per_step"],3), d["pcie"], "enq", round(d["host_enqueue_ms_per_step"],3), d["verified_frames"] and d["verified_frames"]["frames"])
PY
done
