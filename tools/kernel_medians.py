import csv,glob,sys,collections
for d in sys.argv[1:]:
    f=glob.glob(d+'/**/*kernel_trace.csv',recursive=True)[0]
    by=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        by[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    print(d)
    for k,v in by.items():
        v.sort(); print('  %-60s n=%d med=%.1f avg=%.1f'%(k,len(v),v[len(v)//2],sum(v)/len(v)))
