for rep in 1 2; do for v in 0 1; do
echo -n "small_separate=$v C2 full: "; ORBFE_ARUCO_SMALL_SEPARATE=$v python bench.py --cpu-frames 0 --no-verify 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],3), round(b['stage_us_last_step']['aruco_contours']))"
echo -n "small_separate=$v C2 aruco: "; ORBFE_ARUCO_SMALL_SEPARATE=$v python bench.py --cpu-frames 0 --no-verify --no-orb 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],3), round(b['stage_us_last_step']['aruco_contours']))"
echo -n "small_separate=$v C3 full: "; ORBFE_ARUCO_SMALL_SEPARATE=$v python bench.py --cpu-frames 0 --no-verify --config C3 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],3), round(b['stage_us_last_step']['aruco_contours']))"
done; done
ORBFE_ARUCO_SMALL_SEPARATE=1 python -m pytest tests/test_aruco_gpu.py -x -q 2>&1 | tail -2
