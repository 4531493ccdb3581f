# ON THE GPU BOX: phase-lock stages at C3 / C5 with the tiled + banded contour path (2 runs each)
for c in C3 C5; do
for pp in 1 2 3; do for dp in 2 3 4 14; do
  r=""
  for i in 1 2; do
    v=$(ORBFE_PHASE_PIN=$pp ORBFE_DET_PIN=$dp timeout -k 5 200 python bench.py --config $c --cpu-frames 0 --no-verify --no-extras --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])")
    r="$r $v"
  done
  echo "$c phase_pin $pp det_pin $dp : $r"
done; done; done
for c in C3 C5; do
for e in "ORBFE_DEFER_POST=1" "ORBFE_DET_NOFORK=1" "ORBFE_DEFER_POST=1 ORBFE_DET_NOFORK=1" "ORBFE_ENGINE_SETS=1" "ORBFE_ENGINE_SETS=3"; do
  r=""
  for i in 1 2; do
    v=$(env $e timeout -k 5 200 python bench.py --config $c --cpu-frames 0 --no-verify --no-extras --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])")
    r="$r $v"
  done
  echo "$c $e : $r"
done; done
