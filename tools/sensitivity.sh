# Sensitivity of the step time to each kernel's work (diagnosis build): the step with one launch DOUBLED, so that nothing
# downstream changes.  d(step) / (the kernel's time alone) ~ 1: the kernel's work is on the bottleneck; ~ 0: it is hidden.
set -e
cd "$(dirname "$0")/.."
mkdir -p build
( cd orb_slam2_aruco_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -shared -DORBFE_ABLATION -o ../../build/liborbfe_ablate.so orb_kernels.hip orb_extractor.hip match_kernels.hip aruco_kernels.hip \
    aruco_detector.hip bow_vocabulary.hip keyframe_io.hip )
export ORBFE_LIB=$PWD/build/liborbfe_ablate.so
run() { python bench.py --cpu-frames 0 --no-verify --steps 30 ${CFG} 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-34s %.3f ms' % ('$1', d['ms_per_step']))"; }
for rep in 1 2; do
run "as is"
ORBFE_ORB_SKIP=$((1<<8)) run "FAST twice (+400 us alone)"
ORBFE_ORB_SKIP=$((8<<8)) run "blur twice (+275)"
ORBFE_ORB_SKIP=$((4<<8)) run "orient twice (+260)"
ORBFE_ORB_SKIP=$((2<<8)) run "quadtree twice (+85)"
ORBFE_ARUCO_SKIP=$((1<<8)) run "contours twice (+670)"
ORBFE_ARUCO_SKIP=$((2<<8)) run "decode twice (+170)"
ORBFE_ARUCO_SKIP=$((8<<8)) run "threshold twice (+100)"
ORBFE_ORB_SKIP=$((16<<8)) run "resize chain twice (+190)"
ORBFE_MATCH_TWICE=1 run "knn2 + search_init twice (+290)"
done
