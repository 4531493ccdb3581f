# Diagnosis: step time of the concurrent pipeline with individual launches left out (results are invalid while a
# launch is skipped) and with more hardware queues.  Run on the GPU box:  bash tools/ablate.sh
run() { python bench.py --cpu-frames 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s %.3f ms' % ('$1', d['ms_per_step']))"; }
run full
GPU_MAX_HW_QUEUES=8 run "full, 8 hw queues"
GPU_MAX_HW_QUEUES=2 run "full, 2 hw queues"
GPU_MAX_HW_QUEUES=1 run "full, 1 hw queue"
if [ "$1" = "all" ]; then
ORBFE_ARUCO_SKIP=1 run "no contours"
ORBFE_ARUCO_SKIP=2 run "no decode"
ORBFE_ARUCO_SKIP=4 run "no finalize"
ORBFE_ARUCO_SKIP=15 run "aruco = pyramid only"
fi
