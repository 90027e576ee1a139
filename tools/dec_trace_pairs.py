import numpy as np, sys
d=np.fromfile(sys.argv[1],dtype=np.uint64).reshape(-1,32)
n=len(d)
hw,xcc=d[:,0].astype(np.int64),d[:,1].astype(np.int64)
t0=d[:,2].astype(np.int64).min()
start=(d[:,2].astype(np.int64)-t0)/100.; end=(d[:,3].astype(np.int64)-t0)/100.
w=d[:,5:].copy().view(np.uint32).reshape(n,-1)
lo=lambda x: ((x.astype(np.int64)-(t0&0xffffffff))&0xffffffff)/100.
cu=(xcc<<16)|(((hw>>13)&7)<<8)|(((hw>>12)&1)<<4)|((hw>>8)&15)
waveid=hw&15; simd=(hw>>4)&3
print('wave0 wave_id histogram',np.bincount(waveid).tolist(),'simd',np.bincount(simd).tolist())
# pass boundaries per WG: T[p] = end of pass p BN barrier; T[0]= after prologue
T=np.zeros((n,10)); T[:,0]=lo(w[:,0])
for p in range(1,10): T[:,p]=lo(w[:,2*p+1])
C=np.zeros((n,10))
for p in range(1,10): C[:,p]=lo(w[:,2*p])
# for each WG and pass, find mate: WG on same cu overlapping at the pass midpoint
bycu={}
for i in range(n): bycu.setdefault(cu[i],[]).append(i)
rows=[]
for i in range(n):
    for p in range(1,10):
        a,b=T[i,p-1],T[i,p]
        mid=(a+b)/2
        mates=[j for j in bycu[cu[i]] if j!=i and start[j]<mid<end[j]]
        if not mates:
            rows.append((b-a,-1.0,p,start[i]<1.0)); continue
        j=mates[0]
        # mate's phase at our pass start a: find q with T[j,q-1] <= a < T[j,q]
        ph=-2.0
        for q in range(1,10):
            if T[j,q-1]<=a<T[j,q]:
                ph=(a-T[j,q-1])/(T[j,q]-T[j,q-1]); break
        rows.append((b-a,ph,p,start[i]<1.0))
R=np.array(rows,dtype=float)
print('passes with no mate:',(R[:,1]==-1).sum(),'mean dur',R[R[:,1]==-1,0].mean() if (R[:,1]==-1).any() else None)
print('mate in prologue/epilogue (phase -2):',(R[:,1]==-2).sum(), R[R[:,1]==-2,0].mean() if (R[:,1]==-2).any() else None)
ok=R[R[:,1]>=0]
for k in range(10):
    sel=(ok[:,1]>=k/10)&(ok[:,1]<(k+1)/10)
    if sel.any(): print(f'mate phase {k/10:.1f}-{(k+1)/10:.1f}: n={sel.sum():5d} mean pass {ok[sel,0].mean():.2f} us')
for p in range(1,10):
    sel=R[:,2]==p
    print('pass',p,'round1 mean %.2f'%R[sel&(R[:,3]==1),0].mean(),'round2 mean %.2f'%R[sel&(R[:,3]==0),0].mean())
print()
shown=0
for c,idx in bycu.items():
    r1=[i for i in idx if start[i]<1.0]
    if len(r1)!=2: continue
    i,j=r1
    print('CU',hex(c),'WG',i,j)
    for p in range(1,10):
        print(f'  p{p}: A cn {C[i,p]-T[i,p-1]:5.2f} bn {T[i,p]-C[i,p]:5.2f} [{T[i,p-1]:7.2f}..{T[i,p]:7.2f}]   B cn {C[j,p]-T[j,p-1]:5.2f} bn {T[j,p]-C[j,p]:5.2f} [{T[j,p-1]:7.2f}..{T[j,p]:7.2f}]  B-A start offset {T[j,p-1]-T[i,p-1]:6.2f}')
    shown+=1
    if shown>=3: break
