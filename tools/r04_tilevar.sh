# ON THE GPU BOX: build variants of k_ct_walk (tools/tiles_variant.sh) at C5 (tiled by default) and C2 (tiled forced), 2 rounds
for rep in 1 2; do
for v in base steps1 steps3 steps4 take512; do
  lib=orb_slam2_aruco_amd/liborbfe.so; [ $v != base ] && lib=build/liborbfe_$v.so
  a=$(ORBFE_LIB=$PWD/$lib python bench.py --config C5 --cpu-frames 0 --no-verify --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f contours %d alone %d' % (d['ms_per_step'], d['stage_us']['aruco_contours'], d['roofline']['stages']['aruco_contours'].get('launch_us_alone') or 0))")
  b=$(ORBFE_ARUCO_TILED=1 ORBFE_LIB=$PWD/$lib python bench.py --cpu-frames 0 --no-verify --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f contours %d alone %d' % (d['ms_per_step'], d['stage_us']['aruco_contours'], d['roofline']['stages']['aruco_contours'].get('launch_us_alone') or 0))")
  echo "$v  C5: $a   C2 tiled: $b"
done; done
