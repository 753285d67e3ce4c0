#!/bin/bash
# A/B of one environment switch on the default (C2) bench, alternating runs on one box: tools/ab_env.sh NAME A B [bench args]
set -o pipefail
name=$1; a=$2; b=$3; shift 3
for i in 1 2 3; do
  for v in $a $b; do
    env $name=$v timeout -k 10 120 python bench.py --cpu-procs 0 --no-profile --steps 20 --warmup 5 "$@" > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { tail -5 gpurun_out/ab_tmp.err; exit 1; }
    python -c "import json; d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1]); print('$name=$v', d['ms_per_step'], 'ms', d['value'], d['config'].get('objects_in_optimal_catalog'))"
  done
done
