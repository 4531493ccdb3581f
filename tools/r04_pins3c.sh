# ON THE GPU BOX: C3 detector lock stage 4 (default) against 3 and 2, interleaved, 8 rounds
for i in 1 2 3 4 5 6 7 8; do for dp in 4 3 2; do
  v=$(ORBFE_DET_PIN=$dp timeout -k 5 200 python bench.py --config C3 --cpu-frames 0 --no-verify --no-extras --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])")
  echo "det_pin_$dp $v"
done; done | sort | awk '{a[$1]=a[$1]" "$2; s[$1]+=$2; n[$1]++} END {for (k in a) printf "%s mean %.3f :%s\n", k, s[k]/n[k], a[k]}'
