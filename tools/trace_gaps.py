import collections, sys
rows=[]
for line in open(sys.argv[1]):
    q,st,n,s,e=line.rstrip('\n').split('\t'); rows.append((int(s),int(e),n,q,st))
rows.sort()
# main run = after the largest gap in the first half (warm-up/main boundary)
t0=rows[0][0]
prev_end=None; gaps=[]
for i,(s,e,n,q,st) in enumerate(rows):
    if prev_end is not None and s>prev_end: gaps.append((s-prev_end, prev_n, n, (s-t0)/1e6, i))
    if prev_end is None or e>prev_end: prev_end=e; prev_n=n
big=sorted(gaps,reverse=True)[:8]
print("largest gaps (ms, at ms):", [(round(g[0]/1e6,2), round(g[3])) for g in big])
# restrict to the main run: after the largest gap
cut=max(gaps)[4]
main=rows[cut:]
span=(main[-1][1]-main[0][0])/1e6
prev_end=None; g2=[]; busy=0
for s,e,n,q,st in main:
    if prev_end is not None and s>prev_end: g2.append((s-prev_end, prev_n, n))
    if prev_end is None: busy+=e-s; prev_end=e; prev_n=n
    elif e>prev_end: busy+=e-max(s,prev_end); prev_end=e; prev_n=n
print(f"main run: span {span:.1f} ms busy(union) {busy/1e6:.1f} ms idle {sum(g[0] for g in g2)/1e6:.1f} ms kernels {len(main)}")
c=collections.defaultdict(lambda:[0,0])
for g in g2:
    k=(g[1][:28],g[2][:28]); c[k][0]+=1; c[k][1]+=g[0]
for k,v in sorted(c.items(), key=lambda kv:-kv[1][1])[:10]: print(f"  {v[1]/1e6:7.1f} ms {v[0]:5d} x {v[1]/v[0]/1e3:7.1f} us  {k}")
tot=collections.defaultdict(lambda:[0,0])
for s,e,n,q,st in main: tot[n[:45]][0]+=1; tot[n[:45]][1]+=e-s
for k,v in sorted(tot.items(), key=lambda kv:-kv[1][1])[:12]: print(f"{v[1]/1e6:9.1f} ms {v[0]:6d} {v[1]/v[0]/1e3:8.1f} us  {k}")
