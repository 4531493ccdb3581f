for i in 1 2; do
python bench.py --latency --cpu-frames 0 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('latency auto  ', round(d['value'],3), {k: round(v,3) for k,v in d['median_ms'].items()}, 'paired', round(d['paired']['value'],3), {k: round(v,3) for k,v in d['paired']['median_ms'].items()})"
ORBFE_ARUCO_TILED=0 python bench.py --latency --cpu-frames 0 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('latency legacy', round(d['value'],3), {k: round(v,3) for k,v in d['median_ms'].items()}, 'paired', round(d['paired']['value'],3), {k: round(v,3) for k,v in d['paired']['median_ms'].items()})"
done
