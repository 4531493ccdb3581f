# large-frame relay kernels launched in chunks of N frames (their workgroups own a CU's LDS): step and contour stage, C3 and C5
for cfg in C3 C5; do for n in 100000 192 128 96 64; do
echo -n "$cfg chunk $n: "; ORBFE_ARUCO_RELAY_CHUNK=$n python bench.py --cpu-frames 0 --no-verify --config $cfg 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],3), {k:round(v) for k,v in b['stage_us_last_step'].items() if k in ('fast_cells','orient_describe','aruco_contours','resize')})"
done; done
ORBFE_ARUCO_RELAY_CHUNK=64 python -m pytest tests/test_aruco_gpu.py -x -q 2>&1 | tail -1
