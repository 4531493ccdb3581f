mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_aruco_gpu.py -m gpu -x -q -k "tiled_and_one or structured_binary or full_hd or dense_frame" 2>&1 | tail -4
for c in C5 C3; do
  a=$(ORBFE_ARUCO_TILED=1 timeout 300 python bench.py --config $c --cpu-frames 0 --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms contours %d alone %d verified %s' % (d['ms_per_step'], d['stage_us']['aruco_contours'], d['roofline']['stages']['aruco_contours'].get('launch_us_alone') or 0, d['verified_frames'] and d['verified_frames']['frames']))")
  echo "$c tiled  $a"
done
bash tools/kstats.sh gpurun_out/r04c5_alone.csv --config C5 --no-orb --steps 10 > /dev/null 2>&1; python tools/kstat_summary.py gpurun_out/r04c5_alone.csv 6
