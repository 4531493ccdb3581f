# ON THE GPU BOX: band height of k_ct_band at C3 / C5 with the final kernels (records with chain codes, 9 KB of list LDS), 3 interleaved rounds;
# then the rocprofv3 kernel-trace summaries of the C3 / C5 bench commands
run() { c=$1; shift; v=$(env "$@" timeout -k 5 200 python bench.py --config $c --cpu-frames 0 --no-verify --no-extras --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])"); echo "$c $* $v"; }
for rep in 1 2 3; do
for r in 0 3 4 5 6 8; do run C3 ORBFE_ARUCO_BAND_ROWS=$r; done
for r in 0 2 3 4 5 6; do run C5 ORBFE_ARUCO_BAND_ROWS=$r; done
done | sort | awk '{k=$1" "$2; a[k]=a[k]" "$3; s[k]+=$3; n[k]++} END {for (k in a) printf "%s mean %.3f :%s\n", k, s[k]/n[k], a[k]}' | sort
timeout -k 5 200 bash tools/kstats.sh gpurun_out/r04_kernel_stats_c3.csv --config C3 --steps 10 > gpurun_out/r04_kernel_stats_c3.txt 2>&1
timeout -k 5 200 bash tools/kstats.sh gpurun_out/r04_kernel_stats_c5.csv --config C5 --steps 10 > gpurun_out/r04_kernel_stats_c5.txt 2>&1
head -14 gpurun_out/r04_kernel_stats_c5.txt
