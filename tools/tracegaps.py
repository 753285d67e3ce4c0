"""Timeline of the last bench step in a rocprofv3 kernel trace: busy time per queue, idle gaps of the GPU and of the main queue."""
import csv, glob, sys, collections
fn = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Queue_Id'], r['Kernel_Name'][:60]) for r in rows)
starts = [e[0] for e in ev if 'cross_power' in e[3]]
s0 = starts[-(int(sys.argv[2]) if len(sys.argv) > 2 else 1)]       # argv[2]: tiles per step (one cross_power per tile)
sel = [e for e in ev if e[0] >= s0]
span = (max(e[1] for e in sel) - s0) / 1e6
if len(sys.argv) > 3 and sys.argv[3] == 'between':
    # a whole step in the middle of the run, from its first kernel to the next step's first kernel (the idle tail included)
    k = int(sys.argv[2])
    s0, s1 = starts[-2 * k], starts[-k]
    sel = [e for e in ev if s0 <= e[0] < s1]
    span = (s1 - s0) / 1e6
    print('step of %.2f ms: last kernel ends at %.2f ms' % (span, (max(e[1] for e in sel) - s0) / 1e6))
byq = collections.defaultdict(float)
for e in sel:
    byq[e[2]] += (e[1] - e[0]) / 1e6
busy, cs, ce = 0, None, None
gaps = []
last = None
for s, e, q, n in sel:
    if ce is None or s > ce:
        if ce is not None:
            busy += ce - cs
            gaps.append(((s - ce) / 1e3, last[3], last[2], n, q, (s - s0) / 1e6))
        cs, ce, last = s, e, (s, e, q, n)
    elif e > ce:
        ce, last = e, (s, e, q, n)
busy += ce - cs
print('span ms %.2f  busy union ms %.2f  per queue %s' % (span, busy / 1e6, dict(byq)))
mainq = max(byq, key=byq.get)
mq = [e for e in sel if e[2] == mainq]
print('main queue: first start %.2f last end %.2f ms' % ((mq[0][0] - s0) / 1e6, (mq[-1][1] - s0) / 1e6))
gaps.sort(reverse=True)
for g in gaps[:12]:
    print('%.0f us idle after %s(q%s) before %s(q%s) at %.1f ms' % g)
acc = collections.defaultdict(lambda: [0, 0.0])
for a, b in zip(mq, mq[1:]):
    g = (b[0] - a[1]) / 1e3
    if g > 5:
        acc[(a[3][:45], b[3][:45])][0] += 1
        acc[(a[3][:45], b[3][:45])][1] += g
print('main-queue gaps > 5 us: total %.2f ms' % (sum(v[1] for v in acc.values()) / 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
    print('%6.0f us in %3d gaps: %s -> %s' % (v[1], v[0], k[0], k[1]))
