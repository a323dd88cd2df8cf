import csv, sys, glob
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
ev=[]
for r in rows:
    n=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','').split('(')[0]
    ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), n, r['Stream_Id']))
ev.sort()
fills=[e for e in ev if e[2]=='kp_sw_kernel']
a=fills[-4][0]; b=fills[-2][1]
for s,e,n,st in ev:
    if e>a and s<b and (e-s)>400000:
        print(f"{(s-a)/1e6:8.2f} -> {(e-a)/1e6:8.2f}  ({(e-s)/1e6:6.2f} ms)  stream {st:>3}  {n}")
