# build/liborbfe_<name>.so with extra compiler flags:  bash tools/build_variant.sh <name> <flags ...>   (then tools/ab.sh)
# (one object per source, with the per-file flags __graft_entry__.build() uses: the matrix-core filter kernels in the MFMA VGPR form)
N=$1; shift
cd "$(dirname "$0")/../orb_slam2_aruco_amd/csrc" && mkdir -p ../../build/obj_$N || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden"
for f in orb_kernels orb_extractor match_kernels aruco_kernels aruco_tiles aruco_modes aruco_detector bow_vocabulary keyframe_io pipeline; do
  X=""; [ $f = orb_kernels -o $f = aruco_kernels ] && X="-mllvm -amdgpu-mfma-vgpr-form"
  hipcc $F $X "$@" -c $f.hip -o ../../build/obj_$N/$f.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=orbfe.map -o ../../build/liborbfe_$N.so ../../build/obj_$N/*.o && rm -rf ../../build/obj_$N
