# ON THE GPU BOX: the C2 (or "$1" = "--config C3") step of the diagnosis build (-DORBFE_ABLATION) with launches of BOTH chains left out in pairs:
# which chain sets the period, and what shortening both would buy.  bash tools/ablate_pairs.sh ["<bench args>"]
set -e
ARGS=$1   # e.g. "--config C3"
cd $GRAFT_REPO_ROOT
( cd orb_slam2_aruco_amd/csrc && ls *.hip | xargs -P 8 -I{} hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DORBFE_ABLATION -c {} -o /tmp/{}.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/liborbfe_ablate.so /tmp/*.hip.o )
export ORBFE_LIB=$PWD/build/liborbfe_ablate.so
run() { python bench.py --cpu-frames 0 --no-verify --no-extras $ARGS 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %.3f ms' % ('$1', d['ms_per_step']))"; }
for r in 1 2; do
run full
ORBFE_ORB_SKIP=16 run "no resize chain"
ORBFE_ARUCO_SKIP=6 run "no decode + finalize"
ORBFE_ORB_SKIP=16 ORBFE_ARUCO_SKIP=6 run "no resize, no decode + finalize"
ORBFE_ORB_SKIP=16 ORBFE_ARUCO_SKIP=7 run "no resize, no contours/decode/finalize"
ORBFE_ORB_SKIP=2 ORBFE_ARUCO_SKIP=6 run "no quadtree, no decode + finalize"
ORBFE_ORB_SKIP=18 ORBFE_ARUCO_SKIP=6 run "no resize/quadtree, no decode+finalize"
ORBFE_ORB_SKIP=4 run "no orient_describe"
done
