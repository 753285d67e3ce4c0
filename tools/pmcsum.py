import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    if 'copyBuffer' in k: continue
    w=sum(v['SQ_WAVES'])/len(v['SQ_WAVES']) if 'SQ_WAVES' in v else 1
    print(k, 'waves=%d'%w, {c: round(sum(x)/len(x)/w,1) for c,x in v.items() if c!='SQ_WAVES'})
