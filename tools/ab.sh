#!/bin/bash
# A/B on one box: C3-shaped bench with and without an environment switch, alternating (tools/ab.sh VAR)
set -o pipefail
V=$1
for i in $(seq 1 ${REPS:-3}); do
for v in 1 0; do
  export $V=$v
  timeout -k 10 200 python bench.py --cpu-procs 0 --no-profile --steps ${STEPS:-10} --warmup 3 --tile 840x1440 --freqs 2 --tiles-per-gpu 8 --scales 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$v', d['ms_per_step'], d['value'])" || exit 1
done; done
