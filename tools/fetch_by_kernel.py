import csv,glob,collections,re,sys
d=sys.argv[1]; pat=sys.argv[2]
tot=collections.defaultdict(float); cnt=collections.Counter()
for fn in glob.glob(d+'/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(fn)):
        if r['Counter_Name']!='FETCH_SIZE': continue
        n=re.sub(r'\(.*','',r['Kernel_Name'].replace('smg::',''))
        if re.search(pat,n): tot[n]+=float(r['Counter_Value']); cnt[n]+=1
for n in tot: print('%-60s launches %4d  fetch x2 %.1f MB / launch'%(n[:60],cnt[n],tot[n]*2/1e3/cnt[n]))
