#!/bin/bash
# The measurements behind profiles/rNN_*: the default bench line (C2), its rocprofv3 kernel-trace summary, the two PMC passes of the
# same workload (FETCH_SIZE, WRITE_SIZE - separate runs, counters only), then the C3-shaped and the injection-loop lines with their
# kernel summaries.  Everything lands in gpurun_out/ev/; copy what is to be kept into profiles/.
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/ev
rm -rf $O; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err || { tail -5 $O/bench_c2.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_c2 -o c2 -- python bench.py --cpu-procs 0 > $O/bench_c2_traced.json 2> $O/tr_c2.err || { tail -20 $O/tr_c2.err; exit 1; }
cp $O/tr_c2/*kernel_stats.csv $O/bench_c2_kernel_stats.csv 2>/dev/null || cp $O/tr_c2/*/*kernel_stats.csv $O/bench_c2_kernel_stats.csv
find $O/tr_c2 -name '*kernel_trace*' -delete
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o f -- python bench.py --cpu-procs 0 --no-profile --steps 1 --warmup 1 > $O/pmc_f.json 2> $O/pmc_f.err || { tail -20 $O/pmc_f.err; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o w -- python bench.py --cpu-procs 0 --no-profile --steps 1 --warmup 1 > $O/pmc_w.json 2> $O/pmc_w.err || { tail -20 $O/pmc_w.err; exit 1; }
python tools/mkpmc.py $O/pmc_f $O/pmc_w $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1 || { tail -5 $O/pmc_traffic.txt; exit 1; }
rm -rf $O/pmc_f $O/pmc_w
timeout -k 10 300 python bench.py --cpu-procs 0 --tile 840x1440 --freqs 2 --tiles-per-gpu 8 --scales 16 > $O/bench_c3.json 2> $O/bench_c3.err || { tail -5 $O/bench_c3.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_c3 -o c3 -- python bench.py --cpu-procs 0 --no-profile --steps 3 --warmup 2 --tile 840x1440 --freqs 2 --tiles-per-gpu 8 --scales 16 > $O/bench_c3_traced.json 2> $O/tr_c3.err || { tail -20 $O/tr_c3.err; exit 1; }
python tools/tracegaps.py $O/tr_c3 8 > $O/bench_c3_timeline.txt 2>&1 || true
cp $O/tr_c3/*kernel_stats.csv $O/bench_c3_kernel_stats.csv 2>/dev/null || cp $O/tr_c3/*/*kernel_stats.csv $O/bench_c3_kernel_stats.csv
find $O/tr_c3 -name '*kernel_trace*' -delete
timeout -k 10 300 python bench.py --cpu-procs 0 --mode inject --tile 480x3240 --freqs 3 --tiles-per-gpu 4 > $O/bench_inject_c5.json 2> $O/bench_inject.err || { tail -5 $O/bench_inject.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr_inj -o inj -- python bench.py --cpu-procs 0 --no-profile --steps 3 --warmup 2 --mode inject --tile 480x3240 --freqs 3 --tiles-per-gpu 4 > $O/bench_inject_traced.json 2> $O/tr_inj.err || { tail -20 $O/tr_inj.err; exit 1; }
cp $O/tr_inj/*kernel_stats.csv $O/bench_inject_kernel_stats.csv 2>/dev/null || cp $O/tr_inj/*/*kernel_stats.csv $O/bench_inject_kernel_stats.csv
find $O/tr_inj -name '*kernel_trace*' -delete
ls -la $O
