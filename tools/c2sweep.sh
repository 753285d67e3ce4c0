#!/bin/bash
# C2 bench under a few lane / group settings
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/c2sweep.log
: > $O
while read -r envs; do
  [ -z "$envs" ] && continue
  echo "== $envs" >> $O
  env $envs timeout -k 10 200 python bench.py --cpu-procs 0 --no-profile --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json;b=json.loads(sys.stdin.readline());print(b['value'], b['ms_per_step'])" >> $O || { echo failed >> $O; exit 1; }
done <<LIST
NEMO_AMD_SCALE_LANES=2 NEMO_AMD_SCALE_GROUP=2
NEMO_AMD_SCALE_LANES=1 NEMO_AMD_SCALE_GROUP=1
NEMO_AMD_SCALE_LANES=2 NEMO_AMD_SCALE_GROUP=4
NEMO_AMD_SCALE_LANES=3 NEMO_AMD_SCALE_GROUP=3
NEMO_AMD_SCALE_LANES=2 NEMO_AMD_SCALE_GROUP=2
NEMO_AMD_SCALE_LANES=1 NEMO_AMD_SCALE_GROUP=2
NEMO_AMD_SCALE_LANES=2 NEMO_AMD_SCALE_GROUP=4
NEMO_AMD_SCALE_LANES=3 NEMO_AMD_SCALE_GROUP=3
LIST
cat $O
