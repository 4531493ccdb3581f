# ON THE GPU BOX: share of the band kernel's phases (d) segments and (c) small borders: build variants that skip one (results wrong, timing only)
for v in "" skipc skipd; do
for c in C2 C5; do
  L="ORBFE_X=0"; [ -n "$v" ] && L="ORBFE_LIB=$PWD/build/liborbfe_$v.so"
  a=$(env $L ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=1 timeout -k 5 150 python bench.py --config $c --cpu-frames 0 --steps 5 --no-extras --no-verify 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms contours %d alone %d' % (d['ms_per_step'], d['stage_us']['aruco_contours'], d['roofline']['stages']['aruco_contours'].get('launch_us_alone') or 0))")
  echo "$c ${v:-full}  $a"
done; done
