# ON THE GPU BOX: C2 with the banded tiled contour path against the band height (cell rows per band)
for mode in "ORBFE_X=0" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=1 ORBFE_ARUCO_BAND_ROWS=8" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=1 ORBFE_ARUCO_BAND_ROWS=5" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=1 ORBFE_ARUCO_BAND_ROWS=4" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=1 ORBFE_ARUCO_BAND_ROWS=3" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=1 ORBFE_ARUCO_BAND_ROWS=2" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=1 ORBFE_ARUCO_BAND_ROWS=1"; do
  a=$(env $mode timeout 300 python bench.py --config C2 --cpu-frames 0 --steps 20 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms contours %d alone %d verified %s' % (d['ms_per_step'], d['stage_us']['aruco_contours'], d['roofline']['stages']['aruco_contours'].get('launch_us_alone') or 0, d['verified_frames'] and d['verified_frames']['frames']))")
  echo "C2  $a   $mode"
done
ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=1 ORBFE_ARUCO_BAND_ROWS=3 bash tools/kstats.sh gpurun_out/c2b3.csv --config C2 --steps 5 2>&1 | grep "k_ct_\|k_tail"
