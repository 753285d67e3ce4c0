#!/bin/bash
# SQ counter passes over one bench step (C2): where the wave cycles of each kernel go.  tools/sqpmc.sh -> gpurun_out/sq/
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/sq
rm -rf $O; mkdir -p $O
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p1 -o a -- python bench.py --cpu-procs 0 --no-profile --steps 1 --warmup 1 > $O/p1.json 2> $O/p1.err || { tail -20 $O/p1.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_INST_LEVEL_VMEM SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p2 -o b -- python bench.py --cpu-procs 0 --no-profile --steps 1 --warmup 1 > $O/p2.json 2> $O/p2.err || { tail -20 $O/p2.err; exit 1; }
python - <<'PY'
import csv, glob, collections
for d in ('gpurun_out/sq/p1', 'gpurun_out/sq/p2'):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    names = sorted({c for v in agg.values() for c in v})
    print(d)
    print('kernel'.ljust(50), ' '.join(n[3:][:14].rjust(14) for n in names))
    key = 'SQ_WAVE_CYCLES' if 'SQ_WAVE_CYCLES' in names else 'SQ_INSTS_VALU'
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get(key, 0))[:16]:
        print(k.ljust(50), ' '.join(('%.3g' % v.get(n, 0)).rjust(14) for n in names))
PY
rm -rf $O/p1 $O/p2
