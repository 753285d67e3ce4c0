#!/bin/bash
# kernel trace of the default bench (few steps) -> gpurun_out/prof_tr/tr_results.db
set -o pipefail
export TMPDIR=/tmp
rm -rf gpurun_out/prof_tr
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_tr -o tr -- python bench.py --cpu-procs 0 --steps 4 --warmup 2 "$@" > gpurun_out/tr.json 2> gpurun_out/tr.err || { tail -20 gpurun_out/tr.err; exit 1; }
cut -c1-200 gpurun_out/tr.json
