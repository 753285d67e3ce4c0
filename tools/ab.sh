#!/bin/bash
# A/B of an environment switch on the bench: tools/ab.sh VAR A B
set -o pipefail
for v in "$2" "$3" "$2" "$3"; do
  env "$1=$v" timeout -k 10 200 python bench.py --cpu-procs 0 --no-profile > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -5 gpurun_out/ab.err; exit 1; }
  python - "$1=$v" <<PY
import json,sys
d=json.load(open("gpurun_out/ab.json")); print(sys.argv[1], d["value"], d["ms_per_step"])
PY
done
