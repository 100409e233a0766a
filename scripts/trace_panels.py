import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
pan=[r for r in rows if 'k_panel' in r['Kernel_Name']]
n=len(pan)//3
last=pan[-n:]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000 for r in last]
gaps=[(int(last[i+1]['Start_Timestamp'])-int(last[i]['End_Timestamp']))/1000 for i in range(n-1)]
print('n',n,'sum',sum(d),'gapsum',sum(gaps))
print('dur',' '.join('%.1f'%x for x in d))
print('gap',' '.join('%.1f'%x for x in gaps))
