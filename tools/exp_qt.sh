python -m pytest tests/test_orb_gpu.py -x -q 2>&1 | tail -2
bash tools/kstats.sh gpurun_out/s4_orb_stats.csv --no-aruco 2>&1 | grep -i "distribute\|fast_cells\|orient"
for rep in 1 2 3; do python bench.py --cpu-frames 0 --no-verify 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],3), {k:round(v) for k,v in b['stage_us_last_step'].items()})"; done
python bench.py --cpu-frames 0 --config C3 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],3), {k:round(v) for k,v in b['stage_us_last_step'].items()}, b['verified_frames'])"
