#!/bin/bash
# A/B on one box: the C3-shaped bench under values of an environment variable, alternating (tools/ab.sh VAR v1 v2 ...)
set -o pipefail
V=$1; shift
for i in $(seq 1 ${REPS:-3}); do
for v in "$@"; do
  export $V=$v
  timeout -k 10 200 python bench.py --cpu-procs 0 --no-profile --steps ${STEPS:-10} --warmup 3 --tile 840x1440 --freqs 2 --tiles-per-gpu 8 --scales 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$v', d['ms_per_step'], d['value'])" || exit 1
done; done
