#!/bin/bash
# Round-2 measurement: GPU tests, default bench line, kernel trace of the same command.  Run from the repo root on the GPU box.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
timeout -k 10 400 python bench.py > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err || { tail -20 gpurun_out/bench_r02.err; exit 1; }
cat gpurun_out/bench_r02.json
rm -rf gpurun_out/prof_r02
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r02 -o r02 -- python bench.py --cpu-procs 0 > gpurun_out/bench_r02_traced.json 2> gpurun_out/prof_r02.err || { tail -20 gpurun_out/prof_r02.err; exit 1; }
cat gpurun_out/bench_r02_traced.json
find gpurun_out/prof_r02 -name '*kernel_trace*' -size +8M -delete
ls -R gpurun_out/prof_r02 | head -20
