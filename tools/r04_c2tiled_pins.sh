# ON THE GPU BOX: C2 with the tiled + banded contour path under other lock stages (3 runs each); first line = the default path
run() { r=""; for i in 1 2 3; do v=$(env "$@" timeout -k 5 200 python bench.py --config C2 --cpu-frames 0 --no-verify --no-extras --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])"); r="$r $v"; done; echo "$* : $r"; }
run ORBFE_X=0
for pp in 1 2; do for dp in 2 4 14; do for dfp in 0 1; do
run ORBFE_ARUCO_TILED=1 ORBFE_ARUCO_BANDED=1 ORBFE_PHASE_PIN=$pp ORBFE_DET_PIN=$dp ORBFE_DEFER_POST=$dfp
done; done; done
