#!/bin/bash
# solo per-kernel times of nf_noise_map (tools/noisebench.py: plain Gaussian noise, then correlated noise with objects) at 4096^2
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/noiseprof
rm -rf $O; mkdir -p $O
for c in plain corr; do
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/g$c -o g -- python tools/noisebench.py 4096 1 $c > $O/g$c.log 2>&1 || { tail -20 $O/g$c.log; exit 1; }
echo "== $c"; grep "avg us" $O/g$c.log; python - <<PY
import csv,glob
f=glob.glob('$O/g$c/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_noise' in r['Name']: print(r['Name'][:60].ljust(60), r['Calls'].rjust(6), '%8.1f'%(float(r['AverageNs'])/1e3), '%8.1f'%(float(r['MinNs'])/1e3), '%8.1f'%(float(r['MaxNs'])/1e3))
PY
done
find $O -name '*kernel_trace*' -size +8M -delete
