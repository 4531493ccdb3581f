for rep in 1 2 3; do
for cfg in "" "ORBFE_ARUCO_TILED=1" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BAND_ROWS=6"; do
  ms=$(env $cfg timeout 300 python bench.py --config C3 --cpu-frames 0 --no-verify --steps 10 2>/dev/null | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "C3 $ms  ${cfg:-default}"
done
for cfg in "" "ORBFE_ARUCO_TILED=1" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BAND_ROWS=15"; do
  ms=$(env $cfg timeout 300 python bench.py --cpu-frames 0 --no-verify --steps 20 2>/dev/null | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "C2 $ms  ${cfg:-default}"
done
done
