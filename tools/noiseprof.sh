#!/bin/bash
# solo per-kernel times of nf_noise_map on plain Gaussian noise at several sizes
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/noiseprof
rm -rf $O; mkdir -p $O
for n in 4096 4000 2048 1024; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g$n -o g -- python tools/noisebench.py $n > $O/g$n.log 2>&1 || { tail -20 $O/g$n.log; exit 1; }
echo "== $n"; python - <<PY
import csv,glob
f=glob.glob('$O/g$n/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'k_noise' in r['Name']: print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), '%8.1f'%(float(r['AverageNs'])/1e3), '%8.1f'%(float(r['MinNs'])/1e3), '%8.1f'%(float(r['MaxNs'])/1e3))
PY
done
find $O -name '*kernel_trace*' -size +8M -delete
