# build/liborbfe_<name>.so = the current objects with aruco_tiles.hip recompiled under extra flags (run after __graft_entry__.build()):
#   bash tools/tiles_variant.sh steps3 -DCTW_STEPS=3
N=$1; shift
cd "$(dirname "$0")/../orb_slam2_aruco_amd/csrc" && \
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt "$@" -c aruco_tiles.hip -o ../../build/obj/aruco_tiles_$N.o && \
hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/liborbfe_$N.so $(ls ../../build/obj/*.o | grep -v "aruco_tiles") ../../build/obj/aruco_tiles_$N.o
