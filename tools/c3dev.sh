#!/bin/bash
# C3-shaped bench: serial device time per step (survey) and FFT column entries, per env setting given as arguments
set -o pipefail
mkdir -p gpurun_out
for envs in "$@"; do
  env $envs timeout -k 10 200 python bench.py --cpu-procs 0 --tile 840x1440 --freqs 2 --tiles-per-gpu 8 --scales 16 --steps 5 --warmup 2 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.readline()); k=b['kernels']
print('$envs', 'ms/step', b['ms_per_step'], 'device', b['pipeline']['device_ms_per_step'], {n: (k[n]['calls'], k[n]['avg_us']) for n in ('fft_cols_f32','fft_cols_f64','axis_filter_cols','noise_map','fft_rows_f32')})" || exit 1
done
