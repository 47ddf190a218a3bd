import csv,glob,collections,sys
d=sys.argv[1]
rows=list(csv.DictReader(open(glob.glob(d+'/*_kernel_trace.csv')[0])))
agg=collections.OrderedDict()
for r in rows:
    n=r['Kernel_Name'][:44]
    if any(s in n for s in sys.argv[2:]):
        key=(n, r['Grid_Size_X'], r['Grid_Size_Y'], r.get('VGPR_Count',''), r.get('LDS_Block_Size',''))
        x=agg.setdefault(key,[0,0.0])
        x[0]+=1; x[1]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for k,(c,t) in agg.items():
    print(k, "calls",c, "avg_us %.1f"%(t/c))
