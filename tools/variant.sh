# build/liborbfe_<name>.so = the current objects with the given source files recompiled under extra flags (run after __graft_entry__.build()):
#   bash tools/variant.sh prio2 "aruco_kernels" -DORBFE_PRIO_DET_TAIL=2
N=$1; FILES=$2; shift 2
cd "$(dirname "$0")/../orb_slam2_aruco_amd/csrc" || exit 1
EXCL=""
OBJS=""
for f in $FILES; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden "$@" -c $f.hip -o ../../build/obj/${f}__$N.o || exit 1
  OBJS="$OBJS ../../build/obj/${f}__$N.o"; EXCL="$EXCL -e /$f.o\$"
done
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=orbfe.map -o ../../build/liborbfe_$N.so $(ls ../../build/obj/*.o | grep -v "__\|aruco_tiles_" | grep -v $EXCL) $OBJS
