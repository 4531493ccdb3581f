# ON THE GPU BOX: single-frame detector latency against the tile geometry of the contour walk
for mode in "ORBFE_X=0" "ORBFE_ARUCO_TILE_W=128" "ORBFE_ARUCO_TILE_W=96" "ORBFE_ARUCO_TILE_W=64" "ORBFE_ARUCO_TILE_W=32" "ORBFE_ARUCO_BANDED=1 ORBFE_ARUCO_BAND_ROWS=1" "ORBFE_ARUCO_BANDED=1 ORBFE_ARUCO_BAND_ROWS=2" "ORBFE_ARUCO_TILED=0"; do
  a=$(env $mode timeout 300 python bench.py --latency --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k: round(v, 4) for k, v in d['median_ms'].items()})")
  echo "$a   $mode"
done
