# build/liborbfe_<name>.so from the sources of a COMMIT (default HEAD), for an A/B of the working tree against it with tools/ab.sh:
#     bash tools/build_head.sh base [commit]        then  bash tools/ab.sh 4 "" "base=ORBFE_LIB=$PWD/build/liborbfe_base.so"
N=$1; C=${2:-HEAD}
cd "$(dirname "$0")/.." && rm -rf build/src_$N && mkdir -p build/src_$N && git archive $C orb_slam2_aruco_amd/csrc include | tar -x -C build/src_$N && \
cd build/src_$N/orb_slam2_aruco_amd/csrc && \
for f in *.hip; do X=""; [ $f = orb_kernels.hip -o $f = aruco_kernels.hip ] && grep -q k_blur7_mfma orb_kernels.hip && X="-mllvm -amdgpu-mfma-vgpr-form"; \
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fvisibility=hidden $X -c $f -o $f.o & done; wait; \
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=orbfe.map -o ../../../liborbfe_$N.so *.hip.o && cd ../../.. && rm -rf src_$N && ls -la liborbfe_$N.so
