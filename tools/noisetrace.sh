#!/bin/bash
# kernel stats of the noise kernels inside a short bench run
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/noisetrace
rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python bench.py --cpu-procs 0 --no-profile --steps 3 --warmup 1 "$@" > $O/b.json 2> $O/b.err || { tail -20 $O/b.err; exit 1; }
python - <<PY
import csv,glob
f=glob.glob('$O/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k_noise' in r['Name'] or 'finalize' in r['Name']: print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), '%8.1f'%(float(r['AverageNs'])/1e3), '%8.1f'%(float(r['MinNs'])/1e3), '%8.1f'%(float(r['MaxNs'])/1e3))
PY
find $O -name '*kernel_trace*' -size +8M -delete
