# Diagnosis: step time of the concurrent pipeline with individual launches left out (results are invalid while a
# launch is skipped: bench.py prints "value": null and "skips") and with more hardware queues.  The shipped library cannot
# skip work; this script builds the diagnosis library (-DORBFE_ABLATION) first.  Run on the GPU box:  bash tools/ablate.sh [all]
set -e
cd "$(dirname "$0")/.."
mkdir -p build
( cd orb_slam2_aruco_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -shared -DORBFE_ABLATION -o ../../build/liborbfe_ablate.so orb_kernels.hip orb_extractor.hip match_kernels.hip aruco_kernels.hip aruco_tiles.hip aruco_modes.hip pipeline.hip \
    aruco_detector.hip bow_vocabulary.hip keyframe_io.hip )
export ORBFE_LIB=$PWD/build/liborbfe_ablate.so
run() { python bench.py --cpu-frames 0 --no-verify 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-28s %.3f ms' % ('$1', d['ms_per_step']))"; }
run full
GPU_MAX_HW_QUEUES=8 run "full, 8 hw queues"
GPU_MAX_HW_QUEUES=2 run "full, 2 hw queues"
GPU_MAX_HW_QUEUES=1 run "full, 1 hw queue"
if [ "$1" = "all" ]; then
ORBFE_ARUCO_SKIP=1 run "no contours"
ORBFE_ARUCO_SKIP=2 run "no decode"
ORBFE_ARUCO_SKIP=4 run "no finalize"
ORBFE_ARUCO_SKIP=15 run "aruco = pyramid only"
ORBFE_ORB_SKIP=1 run "no FAST"
ORBFE_ORB_SKIP=2 run "no quadtree"
ORBFE_ORB_SKIP=4 run "no orient/describe"
ORBFE_ORB_SKIP=8 run "no blur"
fi
