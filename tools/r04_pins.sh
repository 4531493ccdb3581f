# ON THE GPU BOX: phase-lock stages of the extractor sets / the detector with the persistent general-quadtree launch (4 runs each)
for pp in 1 2 3; do for dp in 2 3 4 14; do
  r=""
  for i in 1 2 3 4; do
    v=$(ORBFE_LIB=$PWD/build/liborbfe_new.so ORBFE_PHASE_PIN=$pp ORBFE_DET_PIN=$dp timeout -k 5 200 python bench.py --cpu-frames 0 --no-verify --no-extras --steps 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % (d['ms_per_step'] if d.get('ms_per_step') else d.get('diagnostic_ms_per_step', 0)))")
    r="$r $v"
  done
  echo "phase_pin $pp det_pin $dp : $r"
done; done
