# k_contours_relay: threads per frame and walk steps between two looks at the work queue (time alone and C2 step)
mkdir -p gpurun_out/t1
for v in base t1024 s1 s3 s4; do
  lib=build/liborbfe_$v.so; [ $v = base ] && lib=orb_slam2_aruco_amd/liborbfe.so
  echo -n "$v alone: "; ORBFE_LIB=$PWD/$lib bash tools/kstats.sh gpurun_out/t1/x.csv --no-orb 2>&1 | grep "k_contours_relay "
done
bash tools/ab.sh "" base=orb_slam2_aruco_amd/liborbfe.so t1024=build/liborbfe_t1024.so s1=build/liborbfe_s1.so s3=build/liborbfe_s3.so s4=build/liborbfe_s4.so 2>&1 | cut -c1-22
