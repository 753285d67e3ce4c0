#!/bin/bash
# C3-shaped bench under a few settings: "VAR=val VAR2=val" per line of stdin-less list below
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/c3sweep.log
: > $O
while read -r envs; do
  [ -z "$envs" ] && continue
  echo "== $envs" >> $O
  env $envs timeout -k 10 200 python bench.py --cpu-procs 0 --no-profile --tile 840x1440 --freqs 2 --tiles-per-gpu 8 --scales 16 --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json;b=json.loads(sys.stdin.readline());print(b['value'], b['ms_per_step'])" >> $O || { echo failed >> $O; exit 1; }
done <<LIST
NEMO_AMD_SCALE_GROUP=16
NEMO_AMD_SCALE_GROUP=8
NEMO_AMD_SCALE_GROUP=4
NEMO_AMD_SCALE_GROUP=16
NEMO_AMD_SCALE_GROUP=8
NEMO_AMD_SCALE_GROUP=4
NEMO_AMD_SCALE_GROUP=8 NEMO_AMD_SCALE_LANES=4
NEMO_AMD_SCALE_GROUP=16 NEMO_AMD_SCALE_LANES=4
LIST
cat $O
