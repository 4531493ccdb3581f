# experiment: stream priorities of the three engines (extractor, detector, matching); bash tools/prio.sh [list of orb,aruco,match triples]
cd "$(dirname "$0")/.."
LIST=${@:-"0,0,0 -1,0,0 -1,0,-1 0,1,0 -1,1,0 -1,1,-1 0,1,1 0,-1,0 1,0,0 1,-1,0 0,-1,-1 1,-1,1 1,0,1"}
for rep in 1 2; do
for p in $LIST; do
  echo -n "prio $p: "
  ORBFE_STREAM_PRIO=$p python bench.py --cpu-frames 0 --no-verify --steps 30 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],3), {k:round(v) for k,v in b['stage_us_last_step'].items()})"
done; done
