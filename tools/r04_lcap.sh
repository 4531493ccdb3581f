# ON THE GPU BOX: list elements k_ct_lists keeps in LDS (its workgroup's LDS request) at C3 / C5
for c in C3 C5; do
for l in 0 4096 8192 12288; do
  a=$(env ORBFE_ARUCO_LCAP=$l timeout -k 5 200 python bench.py --config $c --cpu-frames 0 --steps 10 --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms contours %d alone %d verified %s' % (d['ms_per_step'], d['stage_us']['aruco_contours'], d['roofline']['stages']['aruco_contours'].get('launch_us_alone') or 0, d['verified_frames'] and d['verified_frames']['frames']))")
  echo "$c lcap $l  $a"
done; done
