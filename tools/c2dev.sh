#!/bin/bash
# C2 bench: serial device time per step (survey) and chosen entries, per env setting given as arguments
set -o pipefail
mkdir -p gpurun_out
for envs in "$@"; do
  env $envs timeout -k 10 200 python bench.py --cpu-procs 0 --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.readline()); k=b['kernels']
print('$envs', 'ms/step', b['ms_per_step'], 'device', b['pipeline']['device_ms_per_step'], {n: k[n]['avg_us'] for n in ('axis_filter_cols','axis_filter_rows','fft_cols_f64','fft_cols_f32','noise_map')})" || exit 1
done
