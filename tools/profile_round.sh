# One round's measurements on the GPU box:  bash tools/profile_round.sh <tag>   (e.g. r02)
# bench lines of C2 / C3 / C5, the rocprofv3 kernel-trace summary of the default bench command, the PMC passes (separate runs,
# --kernel-trace only) with the FETCH_SIZE / WRITE_SIZE calibration, and the per-configuration profiles bench.py reads.
# Everything lands in gpurun_out/<tag>/; copy what should be judged into profiles/.
set -x
TAG=${1:-r02}
cd "$(dirname "$0")/.."
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
python bench.py --out $O/${TAG}_bench.json > $O/bench_c2.log 2>&1
python bench.py --config C3 --out $O/${TAG}_bench_c3.json > $O/bench_c3.log 2>&1
python bench.py --latency --out $O/${TAG}_latency.json > $O/latency.log 2>&1
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -- python $OLDPWD/bench.py --cpu-frames 0 --no-verify > $OLDPWD/$O/prof.log 2>&1 )
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/${TAG}_kernel_stats.csv
python tools/pmc.py > $O/${TAG}_pmc_table.txt 2>&1
cp gpurun_out/pmc_summary.json $O/${TAG}_pmc_summary.json
cp gpurun_out/pmc_calibration.json $O/${TAG}_pmc_calibration.json
python tools/make_profiles.py $O/${TAG}_pmc_summary.json $O/${TAG}_pmc_calibration.json C2 $TAG > $O/make_profiles_c2.log 2>&1
python tools/pmc.py --config C3 > $O/${TAG}_pmc_table_c3.txt 2>&1
cp gpurun_out/pmc_summary.json $O/${TAG}_pmc_summary_c3.json
python tools/make_profiles.py $O/${TAG}_pmc_summary_c3.json $O/${TAG}_pmc_calibration.json C3 ${TAG}_c3 > $O/make_profiles_c3.log 2>&1
python tools/pmc.py --config C5 --steps 2 > $O/${TAG}_pmc_table_c5.txt 2>&1
cp gpurun_out/pmc_summary.json $O/${TAG}_pmc_summary_c5.json
python tools/make_profiles.py $O/${TAG}_pmc_summary_c5.json $O/${TAG}_pmc_calibration.json C5 ${TAG}_c5 > $O/make_profiles_c5.log 2>&1
cp profiles/traffic_C2.json profiles/pmc_stage_C2.json profiles/traffic_C3.json profiles/pmc_stage_C3.json profiles/traffic_C5.json profiles/pmc_stage_C5.json $O/
# second pass of the bench lines: now with the freshly written per-configuration traffic / PMC profiles in the roofline record
python bench.py --out $O/${TAG}_bench.json > $O/bench_c2.log 2>&1
python bench.py --config C3 --out $O/${TAG}_bench_c3.json > $O/bench_c3.log 2>&1
python bench.py --config C5 --steps 10 --out $O/${TAG}_bench_c5.json > $O/bench_c5.log 2>&1
python bench.py --from-host --out $O/${TAG}_bench_from_host_C2.json > $O/bench_fh.log 2>&1
# the plain two-rank command (bench.py starts the ranks itself), both ranks on the one GPU over gloo
ORBFE_BENCH_DEVICE=0 ORBFE_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 6 --warmup 2 --cpu-frames 0 --out $O/${TAG}_bench_2ranks_gloo_one_gpu.json > $O/bench_2ranks.log 2>&1
# which HIP stream / hardware queue the RCCL kernels of the gather run on (the library issues them on the matching stream)
( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$O/prof_fg -- python $OLDPWD/bench.py --gpus 1 --force-gather --cpu-frames 0 --no-verify --steps 5 > $OLDPWD/$O/prof_fg.log 2>&1 )
python tools/gather_stream.py $(find $O/prof_fg -name '*kernel_trace.csv' | head -1) > $O/${TAG}_gather_stream_trace.txt 2>&1
rm -rf $O/prof_fg
# the RCCL branch on the hardware that is there (world size 1), and C4 as far as one GPU runs it (8 gloo ranks on device 0, 8 frames each)
python bench.py --gpus 1 --force-gather --cpu-frames 0 --out $O/${TAG}_bench_rccl_world1_force_gather.json > $O/bench_fg.log 2>&1
# ... and with the gather's kernels on a lowest-priority stream of their own (another pool of hardware queues): interleaved with the plain runs
for r in 1 2 3 4; do
  for v in "plain=" "forcegather=" "gatherstream=ORBFE_GATHER_STREAM=1"; do
    n=${v%%=*}; e=${v#*=}; a="--gpus 1 --cpu-frames 0 --no-verify --no-extras"; [ "$n" != plain ] && a="$a --force-gather"
    env $e python bench.py $a 2>/dev/null | python -c "import sys, json; d = [json.loads(l) for l in sys.stdin if l.startswith('{')][-1]; print('$n', round(d['ms_per_step'], 4))"
  done
done | sort | awk '{s[$1] += $2; n[$1]++; a[$1] = a[$1] " " $2} END {for (k in s) printf "%-14s mean %.4f :%s\n", k, s[k] / n[k], a[k]}' > $O/${TAG}_gather_stream_ab.txt 2>&1
PORT=$((20000 + RANDOM % 20000))
ORBFE_BENCH_DEVICE=0 ORBFE_BENCH_BACKEND=gloo OMP_NUM_THREADS=2 python bench.py --gpus 8 --config C4 --frames 8 --steps 3 --warmup 1 --cpu-frames 0 \
    --out $O/${TAG}_bench_c4_8ranks_gloo_one_gpu.json > $O/bench_c4.log 2>&1
python tools/timeline.py > $O/${TAG}_timeline_full.txt 2>&1
bash tools/wave_timing.sh > $O/${TAG}_wave_timing.txt 2>&1
rm -rf $O/prof gpurun_out/pmc gpurun_out/tl
tail -c 400 $O/bench_c2.log
