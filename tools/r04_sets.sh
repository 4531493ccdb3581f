# ON THE GPU BOX: a second detector engine set and 8 hardware queues under the phase locks (interleaved, 4 rounds)
for rep in 1 2 3 4; do
for cfg in "" "ORBFE_ENGINE_SETS_ARUCO=2" "GPU_MAX_HW_QUEUES=8" "ORBFE_ENGINE_SETS_ARUCO=2 GPU_MAX_HW_QUEUES=8" "ORBFE_ENGINE_SETS_ARUCO=2 GPU_MAX_HW_QUEUES=6"; do
  ms=$(env $cfg python bench.py --cpu-frames 0 --no-verify --steps 30 2>/dev/null | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "$ms  ${cfg:-default}"
done; done
