import re,sys
txt=open('/root/repo/gpurun_out/phase/counts.txt').read()
blocks=re.split(r'== stop after (\S+)\n', txt)[1:]
rows=[]
for i in range(0,len(blocks),2):
    n=blocks[i]; d={}
    for m in re.finditer(r'(SQ_\w+)\s+mean\s+([\d.]+)', blocks[i+1]): d[m.group(1)]=float(m.group(2))
    rows.append((n,d))
N=48184
names={0:"setup",1:"p1 table build",2:"p1 clear",3:"p1 vote",4:"p1 evaluate",5:"p1 undo+argmax",6:"p1 diag scan",7:"case select",8:"p2 table",9:"p2 clear",10:"p2 vote",11:"p2 evaluate",12:"p2 undo+argmax",13:"p2 diag scan",14:"combine"}
prev={k:0 for k in rows[0][1]}
print("%-18s %8s %8s %8s %8s %10s"%("phase","VALU","SALU","LDS","BRANCH","wavecyc"))
for n,d in rows:
    if n=='full':
        print("%-18s %8.1f %8.1f %8.1f %8.1f %10.0f  (total per read)"%("full",d['SQ_INSTS_VALU']/N,d['SQ_INSTS_SALU']/N,d['SQ_INSTS_LDS']/N,d['SQ_INSTS_BRANCH']/N,d['SQ_WAVE_CYCLES']/N)); continue
    print("%-18s %8.1f %8.1f %8.1f %8.1f %10.0f"%(names[int(n)],(d['SQ_INSTS_VALU']-prev['SQ_INSTS_VALU'])/N,(d['SQ_INSTS_SALU']-prev['SQ_INSTS_SALU'])/N,(d['SQ_INSTS_LDS']-prev['SQ_INSTS_LDS'])/N,(d['SQ_INSTS_BRANCH']-prev['SQ_INSTS_BRANCH'])/N,(d['SQ_WAVE_CYCLES']-prev['SQ_WAVE_CYCLES'])/N))
    prev=d
