run() { python bench.py --cpu-frames 0 --no-verify $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-46s %.3f ms' % ('$1', d['ms_per_step']))"; }
export ORBFE_BLUR_PLACE=1
for rep in 1 2; do
for q in "" 8; do for D in 1 2; do
  if [ -n "$q" ]; then export GPU_MAX_HW_QUEUES=$q; else unset GPU_MAX_HW_QUEUES; fi
  export ORBFE_ENGINE_SETS=$D
  ARGS=""; run "queues=${q:-4} engine_sets=$D"
done; done; done
export GPU_MAX_HW_QUEUES=8 ORBFE_ENGINE_SETS=2
for place in 0 1 2; do export ORBFE_BLUR_PLACE=$place; ARGS=""; run "8q sets=2 blur_place=$place"; done
export ORBFE_BLUR_PLACE=1
ARGS="--no-aruco"; run "8q sets=2 no-aruco"
ARGS="--no-orb"; run "8q sets=2 no-orb"
export ORBFE_ENGINE_SETS=3; ARGS=""; run "8q sets=3"
export GPU_MAX_HW_QUEUES=16; ARGS=""; run "16q sets=3"
