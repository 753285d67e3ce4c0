import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')][-1]
d=json.loads(l)
print(d['value'], 'Mpix/s', d['ms_per_step'], 'ms/step', d.get('roofline'), d.get('pipeline'))
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['ms_total']): print('  %-18s calls=%4d total_ms=%8.3f avg_us=%8.2f GB/s=%s' % (k, v['calls'], v['ms_total'], v['avg_us'], v['GBps']))
