# ON THE GPU BOX: three extractor sets at C5 / C3 with the lock stages of round 4 (3 runs each)
run() { c=$1; shift; r=""; for i in 1 2 3; do v=$(env "$@" timeout -k 5 200 python bench.py --config $c --cpu-frames 0 --no-verify --no-extras --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])"); r="$r $v"; done; echo "$c $* : $r"; }
run C5 ORBFE_X=0
run C5 ORBFE_ENGINE_SETS=3
run C5 ORBFE_ENGINE_SETS=3 ORBFE_PHASE_PIN=0
run C5 ORBFE_ENGINE_SETS=3 ORBFE_PHASE_PIN=2
run C5 ORBFE_ENGINE_SETS_ARUCO=2
run C5 ORBFE_RECORD_SETS=6
run C3 ORBFE_ENGINE_SETS_ARUCO=2
run C3 ORBFE_RECORD_SETS=6
