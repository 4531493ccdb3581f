#!/bin/bash
cd "$(dirname "$0")/.."
R=$PWD
export TMPDIR=/tmp
timeout -k 5 900 python -m pytest tests/test_pipeline_gpu.py -m gpu -x -q -k "gather or pgm" 2>&1 | tail -15
