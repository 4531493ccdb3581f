# Sensitivity of the step time to each kernel's work (diagnosis build): the step with one launch DOUBLED, so that nothing
# downstream changes.  d(step) / (the kernel's time alone) ~ 1: the kernel's work is on the bottleneck; ~ 0: it is hidden.
# Build the diagnosis library first (HERE, it travels with the snapshot):  bash tools/build_variant.sh ablate -DORBFE_ABLATION
# then on the GPU box:  bash tools/sensitivity.sh
cd "$(dirname "$0")/.."
export ORBFE_LIB=$PWD/build/liborbfe_ablate.so
run() { python bench.py --cpu-frames 0 --no-verify --no-extras --steps 30 ${CFG} 2>/dev/null | python -c "import json,sys; d=[json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('%-34s %.4f ms' % ('$1', d['ms_per_step']))"; }
for rep in 1 2 3; do
run "as is"
ORBFE_ORB_SKIP=$((1<<8)) run "FAST twice (+467 us alone)"
ORBFE_ORB_SKIP=$((8<<8)) run "blur twice (+248)"
ORBFE_ORB_SKIP=$((4<<8)) run "orient twice (+226)"
ORBFE_ORB_SKIP=$((2<<8)) run "quadtree twice (+61)"
ORBFE_ORB_SKIP=$((16<<8)) run "resize chain twice (+204)"
ORBFE_ARUCO_SKIP=$((1<<8)) run "contours twice (+410)"
ORBFE_ARUCO_SKIP=$((2<<8)) run "decode twice (+116)"
ORBFE_ARUCO_SKIP=$((8<<8)) run "threshold twice (+92)"
done | sort | awk '{k=$0; sub(/ [0-9.]+ ms$/,"",k); v=$(NF-1); s[k]+=v; n[k]++; a[k]=a[k]" "v} END {for (k in s) printf "%-36s mean %.4f :%s\n", k, s[k]/n[k], a[k]}' | sort
