#!/bin/bash
# ON THE GPU BOX: shader clock and power while bench.py runs a long timed region (is the step power- or clock-limited?)
cd "$(dirname "$0")/.."
python bench.py --cpu-frames 0 --no-verify --steps 30000 $@ > /tmp/clk_bench.json 2>/dev/null &
BP=$!
sleep 20
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use|fclk" | tr -s ' ' | head -8
  echo ---
  sleep 0.5
done
wait $BP
python -c "import json; d=json.load(open('/tmp/clk_bench.json')); print('ms_per_step', d['ms_per_step'])"
echo "idle:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | head -4
