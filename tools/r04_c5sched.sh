# ON THE GPU BOX: scheduling switches at C5 / C3 now that the contour stage no longer holds whole CUs (3 interleaved rounds)
for rep in 1 2 3; do
for cfg in "" "ORBFE_ENGINE_SETS=2" "ORBFE_DEFER_POST=1" "ORBFE_DET_NOFORK=1" "ORBFE_ENGINE_SETS=2 ORBFE_DEFER_POST=1"; do
  ms=$(env $cfg timeout 300 python bench.py --config C5 --cpu-frames 0 --no-verify --steps 10 2>/dev/null | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "C5 $ms  ${cfg:-default}"
done
for cfg in "" "ORBFE_ARUCO_TILED=1" "ORBFE_ARUCO_TILED=1 ORBFE_DET_PIN=0" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BAND_ROWS=6"; do
  ms=$(env $cfg timeout 300 python bench.py --config C3 --cpu-frames 0 --no-verify --steps 10 2>/dev/null | python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "C3 $ms  ${cfg:-default}"
done
done
