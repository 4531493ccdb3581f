# ON THE GPU BOX: where a wave's life goes (shader clocks per phase, averaged over all waves of the timed steps), in the pipeline and alone:
#     bash tools/wave_timing.sh [bench args]
# Instrumented build (-DORBFE_WAVE_TIMING: WT_MARK in k_orient_describe2 and k_fast_cells), never shipped.
cd "$(dirname "$0")/.."
bash tools/build_variant.sh wt -DORBFE_WAVE_TIMING > /dev/null 2>&1
for a in "" "--no-aruco"; do
  echo "== bench $a $*"
  ORBFE_LIB=$PWD/build/liborbfe_wt.so ORBFE_BENCH_WAVE_TIMING=1 python bench.py --cpu-frames 0 --no-verify --no-extras $a "$@" 2>&1 >/dev/null | grep wave_timing
done
