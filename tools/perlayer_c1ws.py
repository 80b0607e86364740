import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
ws=[r for r in rows if 'conv1x1_fwd_ws' in r['Kernel_Name']]
ws.sort(key=lambda r:int(r['Start_Timestamp']))
per=len(ws)//5
step=ws[per*2:per*3]
print(' '.join('%.1f'%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in step), 'sum %.1f'%sum((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in step))
