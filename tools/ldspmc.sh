#!/bin/bash
# LDS counters of the line kernels (one PMC pass, SQ block only)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/ldspmc
rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O -o l -- python tools/axisbench.py > $O/run.log 2>&1 || { tail -20 $O/run.log; exit 1; }
python - <<PY
import csv,glob,collections
f=glob.glob('$O/**/*counter_collection.csv', recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    if 'k_fft' in k or 'k_line' in k:
        print(k.ljust(70), {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
