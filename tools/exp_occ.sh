cd "$(dirname "$0")/.."
run() { echo -n "$1: "; env $1 python bench.py --cpu-frames 0 --no-verify --steps 30 $2 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(b['ms_per_step'],3), {k:round(v) for k,v in b['stage_us_last_step'].items()})"; }
for rep in 1 2; do
run "X=0"
run "ORBFE_OCC_FAST=4"
run "ORBFE_OCC_FAST=6"
run "ORBFE_OCC_FAST=3"
run "ORBFE_OCC_FAST=4 ORBFE_OCC_BLUR=4 ORBFE_OCC_ORIENT=4"
run "ORBFE_OCC_FAST=4 ORBFE_OCC_BLUR=2 ORBFE_OCC_ORIENT=4"
run "ORBFE_OCC_FAST=6 ORBFE_OCC_BLUR=4 ORBFE_OCC_ORIENT=6"
done
run "X=0" --no-aruco
run "ORBFE_OCC_FAST=4" --no-aruco
run "ORBFE_OCC_FAST=4 ORBFE_OCC_BLUR=4 ORBFE_OCC_ORIENT=4" --no-aruco
