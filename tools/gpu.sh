#!/bin/bash
# tools/gpu.sh [--timeout S] -- '<command>'   (run HERE, in the build container)
# gpurun with the snapshot stamped: build/COMMIT = the commit (+ "-dirty") the snapshot was taken from, so that profiles made
# on the GPU box (tools/make_profiles.py) can say which source they describe -- .git does not travel.
cd "$(dirname "$0")/.."
mkdir -p build
c=$(git rev-parse --short=12 HEAD)
git diff --quiet HEAD -- . || c="$c-dirty"
echo "$c" > build/COMMIT
exec /usr/local/graft/bin/gpurun "$@"
