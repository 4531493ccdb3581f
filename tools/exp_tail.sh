# k_tail_approx: persistent workgroups (-DRT_WGS) and the length below which borders go four to a wave (-DRT_QUAD_MAX); build the
# variants first:  bash tools/build_variant.sh w1024 -DRT_WGS=1024   etc.
mkdir -p gpurun_out/t1
for cfg in C2 C3; do
for v in base w1024 w2048 q96 q256; do
  lib=build/liborbfe_$v.so; [ $v = base ] && lib=orb_slam2_aruco_amd/liborbfe.so
  echo -n "$cfg $v: "; ORBFE_LIB=$PWD/$lib bash tools/kstats.sh gpurun_out/t1/x.csv --no-orb --config $cfg 2>&1 | grep "k_tail_approx"
done
done
