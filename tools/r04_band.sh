# ON THE GPU BOX: the banded walk kernel (k_ct_band) -- contour tests with the tiled path forced, then C2 / C3 / C5 alone and in the pipeline
mkdir -p gpurun_out
ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=1 timeout 600 python -m pytest tests/test_aruco_gpu.py -m gpu -x -q -k "structured_binary or relay_and_legacy or dense_frame or detect_matches_oracle or lds_boundary or tail_kernel or full_hd" 2>&1 | tail -4
for c in C2 C3 C5; do
for mode in "ORBFE_ARUCO_TILED=0" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=0" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=1" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=1 ORBFE_ARUCO_BAND_ROWS=4"; do
  a=$(env $mode timeout 300 python bench.py --config $c --cpu-frames 0 --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms contours %d alone %d verified %s' % (d['ms_per_step'], d['stage_us']['aruco_contours'], d['roofline']['stages']['aruco_contours'].get('launch_us_alone') or 0, d['verified_frames'] and d['verified_frames']['frames']))")
  echo "$c  $a   $mode"
done; done
