# build/liborbfe_<name>.so with extra compiler flags:  bash tools/build_variant.sh <name> <flags ...>   (then tools/ab.sh)
N=$1; shift
cd "$(dirname "$0")/../orb_slam2_aruco_amd/csrc" && mkdir -p ../../build && \
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden -shared -Wl,--version-script=orbfe.map "$@" \
  -o ../../build/liborbfe_$N.so orb_kernels.hip orb_extractor.hip match_kernels.hip aruco_kernels.hip aruco_tiles.hip aruco_modes.hip aruco_detector.hip bow_vocabulary.hip keyframe_io.hip pipeline.hip
