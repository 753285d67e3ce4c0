#!/bin/bash
# C3-shaped bench under a few settings of the line-transform knobs (rows per workgroup / threads per row / column tile LDS):
# device ms per step and the row / column entry points, one line per setting -> gpurun_out/knobs_c3.txt
set -o pipefail
out=gpurun_out/knobs_c3.txt
: > $out
run() {
    label="$1"; shift
    env "$@" timeout -k 10 120 python bench.py --cpu-procs 0 --steps 6 --warmup 3 --tile 840x1440 --freqs 2 --tiles-per-gpu 8 --scales 16 > gpurun_out/knob_tmp.json 2> gpurun_out/knob_tmp.err || { echo "$label FAILED" >> $out; tail -3 gpurun_out/knob_tmp.err >> $out; return 1; }
    python - "$label" >> $out <<'P'
import json, sys
d = json.loads(open('gpurun_out/knob_tmp.json').read().strip().splitlines()[-1])
k = d['kernels']
print("%-34s step %6.2f ms  device %6.2f ms | " % (sys.argv[1], d['ms_per_step'], d['pipeline']['device_ms_per_step']) +
      "  ".join("%s %.0f" % (n.replace('axis_filter', 'af').replace('fft_', ''), k[n]['avg_us']) for n in
                ('fft_rows_f64', 'fft_rows_f32', 'axis_filter_rows', 'fft_cols_f64', 'fft_cols_f32', 'axis_filter_cols')))
P
}
run "default" NF_DUMMY=0 &&
run "ROW_TB=64" NF_FFT_ROW_TB=64 &&
run "ROW_TB=128" NF_FFT_ROW_TB=128 &&
run "ROW_TB=256" NF_FFT_ROW_TB=256 &&
run "ROWS_PER_WG=2 ROW_LDS=64" NF_FFT_ROW_LDS_KB=64 NF_FFT_ROWS_PER_WG=2 &&
run "ROWS_PER_WG=2 ROW_TB=128" NF_FFT_ROW_LDS_KB=64 NF_FFT_ROWS_PER_WG=2 NF_FFT_ROW_TB=128 &&
run "COL_LDS=16" NF_FFT_COL_LDS_KB=16 &&
run "COL_LDS=64" NF_FFT_COL_LDS_KB=64 &&
run "COL_LDS=64 COL_THREADS=512" NF_FFT_COL_LDS_KB=64 NF_FFT_COL_THREADS=512 &&
run "COL_LDS=128 COL_THREADS=512" NF_FFT_COL_LDS_KB=128 NF_FFT_COL_THREADS=512
cat $out
