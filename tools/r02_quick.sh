#!/bin/bash
# GPU suite + C2 and C3 bench lines (no profiles).  Run from the repo root on the GPU box.
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/quick
rm -rf $O; mkdir -p $O
if [ "$1" != "notests" ]; then
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -40 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
fi
timeout -k 10 300 python bench.py --cpu-procs 0 > $O/c2.json 2> $O/c2.err || { tail -20 $O/c2.err; exit 1; }
python -c "import json;b=json.load(open('$O/c2.json'));print('C2', b['value'], b['ms_per_step'], b['pipeline']['device_ms_per_step'])"
timeout -k 10 300 python bench.py --cpu-procs 0 --tile 840x1440 --freqs 2 --tiles-per-gpu 8 --scales 16 > $O/c3.json 2> $O/c3.err || { tail -20 $O/c3.err; exit 1; }
python -c "import json;b=json.load(open('$O/c3.json'));print('C3', b['value'], b['ms_per_step'], b['pipeline']['device_ms_per_step'])"
