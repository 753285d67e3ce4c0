#!/bin/bash
# kernel trace of the injection-loop bench (few steps): per-kernel stats
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/prof_inj
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_inj -o inj -- python bench.py --cpu-procs 0 --no-profile --steps 3 --warmup 2 --mode inject --tile 480x3240 --freqs 3 --tiles-per-gpu 4 > gpurun_out/injtr.json 2> gpurun_out/injtr.err || { tail -20 gpurun_out/injtr.err; exit 1; }
cut -c1-200 gpurun_out/injtr.json
head -60 gpurun_out/prof_inj/*kernel_stats.csv | cut -c1-170 > gpurun_out/inj_kstats.txt
python tools/tracegaps.py gpurun_out/prof_inj 4 > gpurun_out/inj_gaps.txt 2>&1 || true
find gpurun_out/prof_inj -name '*kernel_trace*' -size +8M -delete
cat gpurun_out/inj_kstats.txt
