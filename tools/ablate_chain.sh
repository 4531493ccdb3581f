# Which launches of the extractor chain set the step?  Diagnosis build; the launches are left out AFTER the warm-up steps, so that
# every buffer downstream holds real data (a skipped resize chain otherwise feeds FAST flat images).  bash tools/ablate_chain.sh
cd "$(dirname "$0")/.."
export ORBFE_LIB=$PWD/build/liborbfe_ablate.so ORBFE_SKIP_AFTER_WARMUP=1
run() { python bench.py --cpu-frames 0 --no-verify --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-34s %.3f ms' % ('$1', d['ms_per_step']))"; }
for rep in 1 2; do
run "as is"
ORBFE_ORB_SKIP=16 run "no resize chain"
ORBFE_ORB_SKIP=2 run "no quadtree"
ORBFE_ORB_SKIP=18 run "no resize chain, no quadtree"
ORBFE_ORB_SKIP=8 run "no blur"
ORBFE_ORB_SKIP=4 run "no orient / describe"
ORBFE_ARUCO_SKIP=1 run "no contours"
done
