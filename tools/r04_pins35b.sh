# ON THE GPU BOX: confirmation runs of the candidates of r04_pins35.sh
run() { c=$1; shift; r=""; for i in 1 2 3 4; do v=$(env "$@" timeout -k 5 200 python bench.py --config $c --cpu-frames 0 --no-verify --no-extras --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])"); r="$r $v"; done; echo "$c $* : $r"; }
run C5 ORBFE_PHASE_PIN=2 ORBFE_DET_PIN=4
run C5 ORBFE_PHASE_PIN=1 ORBFE_DET_PIN=4
run C5 ORBFE_PHASE_PIN=1 ORBFE_DET_PIN=2
run C5 ORBFE_PHASE_PIN=0 ORBFE_DET_PIN=4
run C5 ORBFE_PHASE_PIN=1 ORBFE_DET_PIN=4 ORBFE_DEFER_POST=1
run C5 ORBFE_PHASE_PIN=1 ORBFE_DET_PIN=0
run C3 ORBFE_PHASE_PIN=2 ORBFE_DET_PIN=4
run C3 ORBFE_PHASE_PIN=2 ORBFE_DET_PIN=3
run C3 ORBFE_PHASE_PIN=1 ORBFE_DET_PIN=4
run C3 ORBFE_PHASE_PIN=1 ORBFE_DET_PIN=2
run C3 ORBFE_PHASE_PIN=0 ORBFE_DET_PIN=4
