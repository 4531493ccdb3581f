cd "$(dirname "$0")/.."
for rep in 1 2 3; do
for cfg in "X=0" "ORBFE_NO_LEND=1 GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8" "ORBFE_NO_LEND=1 GPU_MAX_HW_QUEUES=6" "ORBFE_NO_LEND=1"; do
  echo -n "$cfg: "
  env $cfg python bench.py --cpu-frames 0 --no-verify --steps 30 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],3), {k:round(v) for k,v in b['stage_us_last_step'].items()})" | cut -c1-150
done; done
