cd "$(dirname "$0")/.."
for rep in 1 2; do
for cfg in "ORBFE_ENGINE_SETS=1 ORBFE_ENGINE_SETS_ARUCO=1" "ORBFE_ENGINE_SETS=2 ORBFE_ENGINE_SETS_ARUCO=1" "ORBFE_ENGINE_SETS=1 ORBFE_ENGINE_SETS_ARUCO=2" "ORBFE_ENGINE_SETS=2 ORBFE_ENGINE_SETS_ARUCO=2" "ORBFE_ENGINE_SETS=2 ORBFE_ENGINE_SETS_ARUCO=2 GPU_MAX_HW_QUEUES=8" "ORBFE_ENGINE_SETS=1 ORBFE_ENGINE_SETS_ARUCO=2 GPU_MAX_HW_QUEUES=8"; do
  echo -n "$cfg: "
  env $cfg python bench.py --cpu-frames 0 --no-verify --steps 30 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],3), {k:round(v) for k,v in b['stage_us_last_step'].items()})"
done; done
