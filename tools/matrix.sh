run() { python bench.py --cpu-frames 0 --no-verify $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-46s %.3f ms' % ('$1', d['ms_per_step']))"; }
for rep in 1 2; do
for q in "" 8; do for cfg in "1 1" "2 1" "1 2"; do
  set -- $cfg
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  export ORBFE_ENGINE_SETS=$1 ORBFE_ENGINE_SETS_ARUCO=$2
  ARGS=""; run "queues=${q:-4} orb_sets=$1 aruco_sets=$2"
done; done; done
