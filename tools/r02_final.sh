#!/bin/bash
# Round-2 evidence: GPU tests, the two PMC passes (one counter each), default bench line, kernel trace of the same command, C3-shaped line.
# Run from the repo root on the GPU box; every GPU step has its own time limit and the script stops at the first failure.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r02f
rm -rf $O; mkdir -p $O
if [ "$1" != "notests" ]; then
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
fi
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python bench.py --steps 1 --warmup 1 --cpu-procs 0 --no-profile > $O/pmc_fetch.json 2> $O/pmc_fetch.err || { tail -20 $O/pmc_fetch.err; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python bench.py --steps 1 --warmup 1 --cpu-procs 0 --no-profile > $O/pmc_write.json 2> $O/pmc_write.err || { tail -20 $O/pmc_write.err; exit 1; }
python tools/mkpmc.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json > $O/mkpmc.log 2>&1 || { tail $O/mkpmc.log; exit 1; }
cp $O/pmc_traffic.json profiles/r02_pmc_traffic.json      # the bench line below quotes it (roofline.traffic)
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r02 -- python bench.py --cpu-procs 0 > $O/bench_traced.json 2> $O/trace.err || { tail -20 $O/trace.err; exit 1; }
find $O/trace -name '*kernel_trace*' -size +8M -delete
timeout -k 10 300 python bench.py --cpu-procs 0 --tile 840x1440 --freqs 2 --tiles-per-gpu 8 --scales 16 > $O/bench_c3.json 2> $O/bench_c3.err || { tail -20 $O/bench_c3.err; exit 1; }
find $O -name '*counter_collection*' -size +8M -delete
ls -R $O | head -40
