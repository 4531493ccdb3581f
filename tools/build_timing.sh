# Instrumented diagnosis builds (never shipped): phase timers inside the contour kernels (-DORBFE_CT_TIMING) and the quadtree
# kernel (-DORBFE_QT_TIMING).  Run on the GPU box, then e.g.  ORBFE_LIB=build/liborbfe_timing.so python tools/ct_timing.py
cd "$(dirname "$0")/.."
mkdir -p build
cd orb_slam2_aruco_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -shared"
S="orb_kernels.hip orb_extractor.hip match_kernels.hip aruco_kernels.hip aruco_tiles.hip aruco_modes.hip aruco_detector.hip bow_vocabulary.hip keyframe_io.hip pipeline.hip"
hipcc $F -DORBFE_CT_TIMING -DORBFE_QT_TIMING -DORBFE_SFI_TIMING "$@" -o ../../build/liborbfe_timing${TIMING_SUFFIX}.so $S
