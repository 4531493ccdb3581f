# ON THE GPU BOX: chain codes recorded by the walk kernels (k_ct_points decodes) -- contour tests on every tiled variant, then C2 / C3 / C5 and latency
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_aruco_gpu.py tests/test_stress_gpu.py tests/test_natural_images.py -m gpu -x -q 2>&1 | tail -6
for c in C2 C3 C5; do
for mode in "ORBFE_X=0" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=0" "ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=1"; do
  a=$(env $mode timeout 300 python bench.py --config $c --cpu-frames 0 --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms contours %d alone %d verified %s' % (d['ms_per_step'], d['stage_us']['aruco_contours'], d['roofline']['stages']['aruco_contours'].get('launch_us_alone') or 0, d['verified_frames'] and d['verified_frames']['frames']))")
  echo "$c  $a   $mode"
done; done
timeout 300 python bench.py --latency --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({k: d[k] for k in d if 'ms' in k or 'lat' in k})"
bash tools/kstats.sh gpurun_out/codesC5.csv --config C5 --steps 5 2>&1 | grep "k_ct_\|k_tail"
ORBFE_ARUCO_TILED=1 bash tools/kstats.sh gpurun_out/codesC2t.csv --config C2 --steps 5 2>&1 | grep "k_ct_\|k_tail" 
