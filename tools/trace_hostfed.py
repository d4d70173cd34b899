import csv, glob, sys
d=sys.argv[1]
k=[r for r in csv.DictReader(open(glob.glob(d+'/**/*kernel_trace.csv',recursive=True)[0]))]
m=[r for r in csv.DictReader(open(glob.glob(d+'/**/*memory_copy_trace.csv',recursive=True)[0]))]
print('kernels',len(k),'copies',len(m)); print(m[0].keys())
ev=[]
for r in k:
    n=r['Kernel_Name']
    if 'k_front' in n or 'k_back' in n: ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'F' if 'k_front' in n else 'B'))
for r in m:
    b=int(r.get('Bytes') or r.get('Size') or 0) if (r.get('Bytes') or r.get('Size')) else 0
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'copy s'+r['Stream_Id']))
ev.sort()
# take a window in the middle of the first loop
mid=[e for e in ev if e[2]=='B']
t0=mid[100][0]
for e in ev:
    if t0<=e[0]<t0+600000: print(f"{(e[0]-t0)/1e3:8.1f} {(e[1]-t0)/1e3:8.1f} dur {(e[1]-e[0])/1e3:6.1f} {e[2]}")
