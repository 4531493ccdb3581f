mkdir -p gpurun_out/t1
python -m pytest tests/test_aruco_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -2; python tools/stress_aruco.py 2>&1 | tail -1
for cfg in C2 C3 C5; do echo "$cfg:"; bash tools/kstats.sh gpurun_out/t1/x.csv --no-orb --config $cfg 2>&1 | grep "k_tail_"; done
python bench.py 2>&1 | tail -1 | cut -c1-200
