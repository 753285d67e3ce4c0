#!/bin/bash
# kernel trace of the C3-shaped bench (few steps) and the timeline of its last step
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/prof_c3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3 -o c3 -- python bench.py --cpu-procs 0 --no-profile --steps 3 --warmup 2 --tile 840x1440 --freqs 2 --tiles-per-gpu 8 --scales 16 > gpurun_out/c3tr.json 2> gpurun_out/c3tr.err || { tail -20 gpurun_out/c3tr.err; exit 1; }
cut -c1-200 gpurun_out/c3tr.json
python tools/tracegaps.py gpurun_out/prof_c3 8 > gpurun_out/c3_gaps.txt 2>&1 || true
head -40 gpurun_out/c3_gaps.txt
head -45 gpurun_out/prof_c3/*kernel_stats.csv | cut -c1-150 > gpurun_out/c3_kstats.txt
find gpurun_out/prof_c3 -name '*kernel_trace*' -size +8M -delete
