#!/usr/bin/env python3
"""LDS bank-conflict and merge-round model of relax_band_kernel's direct-index merge (kernels_relaxb.h, MpcRbWinAsm), on real
posteriors from the CPU oracle: for one 8 x 8 tile of a synthetic family it lays the step's pieces out as the kernel does
(first pieces, descriptor pieces, value areas), walks the cells in a given order and counts, per (wave, slot): merge rounds
(longest X row among the 64 cells) and LDS cycles of the descriptor read and the value look-ups (ds_read_b32: two groups of 32
lanes, 32 banks, one cycle per distinct address on the busiest bank; MI355X_MICROARCH.md, LDS). It is what picked the cell order
(blocks of 8 rows) and what ruled out two layout ideas before any device time was spent on them:
  order=pair: 2.08 rounds, 16.3 b32 cycles per slot; order=1 (row by row): 1.44 rounds, 24.0; order=8: 1.62 rounds, 20.1;
  misses reading one common zero word: -1.2 .. -2.5 cycles; descriptors of the 8 Y records interleaved: -0.7.
The device agrees in direction and size (profiles/r09b_order_sweep.log: SQ_LDS_BANK_CONFLICT 1.0e11 -> 2.3e11 for order=1).
Test infrastructure: uses the oracle (tests/_parity.py); nothing here is on the product path.
usage: python tests/analysis/lds_conflict_model.py [n_seqs=40] [length=400]"""
import sys, time, numpy as np
sys.path.insert(0, __file__.rsplit('/tests/', 1)[0]); sys.path.insert(0, __file__.rsplit('/tests/', 1)[0] + '/tests')
import _parity as P
from muscle_amd.synth import make_family
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
L = int(sys.argv[2]) if len(sys.argv) > 2 else 400
seqs = make_family(n, L, seed=1)
stages, ea = P.run_oracle(seqs, iters=0, threads=2)
pairs = stages[0]
def pidx(x,y): return x*n - x*(x+1)//2 + (y-x-1)
_cache={}
def rows_of(A,Z):
    """list over rows of A: sorted cols in Z"""
    key=(A,Z)
    if key in _cache: return _cache[key]
    LA=len(seqs[A])
    if A<Z:
        o,v=pairs[pidx(A,Z)]; cols=v[1::2]
        r=[cols[o[i]:o[i+1]].astype(np.int64) for i in range(LA)]
    else:
        o,v=pairs[pidx(Z,A)]; cols=v[1::2].astype(np.int64)
        rws=np.repeat(np.arange(len(o)-1),np.diff(o.astype(np.int64)))
        r=[[] for _ in range(LA)]
        for rw,c in zip(rws,cols): r[c].append(rw)
        r=[np.array(sorted(t),dtype=np.int64) for t in r]
    _cache[key]=r; return r
r0,r1=96,200
XS=list(range(0,8)); YS=list(range(8,16))
# cells per X group
groups=[]
for X in XS:
    cells=[]
    for j,Y in enumerate(YS):
        o,v=pairs[pidx(X,Y)]
        for x in range(r0,min(r1,len(seqs[X]))):
            for e in range(o[x],o[x+1]): cells.append((x,j,int(v[2*e+1])))
    groups.append(cells)
ylo=[min(c[2] for g in groups for c in g if c[1]==j) for j in range(8)]
yhi=[max(c[2] for g in groups for c in g if c[1]==j)+1 for j in range(8)]
def conflicts(addr_lists):
    """addr_lists: list of word addresses (or None inactive) for 64 lanes; b32: groups of 32, 32 banks -> (cycles, extra)"""
    tot=0; extra=0
    for h in (0,32):
        a=[x for x in addr_lists[h:h+32] if x is not None]
        if not a: continue
        banks={}
        for w in set(a): banks.setdefault(w%32,0); banks[w%32]+=1
        m=max(banks.values()); tot+=m; extra+=m-1
    return tot,extra
def run(order,miss_common=False,desc_inter=False,zs=range(0,n,3),rowlen=lambda sp:sp+1):
    res=dict(desc=[0,0],look=[0,0],look2=[0,0],slots=0,iters=0)
    for Z in zs:
        if Z in XS or Z in YS: continue
        # layout
        Xrows=[rows_of(X,Z) for X in XS]; Yrows=[rows_of(Y,Z) for Y in YS]
        fst=0; 
        fst+=8*(r1-r0)
        D=[]; 
        for j in range(8):
            arow=ylo[j]&~3
            D.append((4*fst,arow)); fst+=((yhi[j]+1-arow)+3)//4
        O=fst
        for i in range(8):
            O+=sum(max(0,(len(Xrows[i][x])+1)//2-1) for x in range(r0,min(r1,len(Xrows[i]))))
        V=[]; offs=[]
        for j in range(8):
            yr=Yrows[j]; span=[ (int(r[-1]-r[0]+1) if len(r) else 0) for r in yr]
            off=np.concatenate([[0],np.cumsum([rowlen(s) for s in span])])
            vb0=off[ylo[j]]//4
            V.append(4*O-4*vb0); offs.append((off,span))
            O+= (off[min(yhi[j],len(yr))]+3)//4 - vb0 + 1
        for gi,cells in enumerate(groups):
            if order=="pair": cs=sorted(cells,key=lambda c:(c[1],c[0],c[2]))
            else:
                G=order; cs=sorted(cells,key=lambda c:(c[0]//G,c[1],c[0],c[2]))
            xr=Xrows[gi]
            for s in range(0,len(cs),64):
                w=cs[s:s+64]; w=w+[w[-1]]*(64-len(w))
                da=[];db=[];l0=[];l1=[];m0=[];m1=[]; it=1
                for (x,j,y) in w:
                    if desc_inter: a=8*(y-ylo[j])+j; da.append(a); db.append(a+8)
                    else: a=D[j][0]+(y-D[j][1]); da.append(a); db.append(a+1)
                    off,span=offs[j]; yr=Yrows[j][y]; c0=int(yr[0]) if len(yr) else 0; sp=span[y]
                    ent=xr[x]
                    def look(z):
                        jx=z-c0
                        if jx<0 or jx>=sp:
                            return -1 if miss_common else V[j]+int(off[y])+sp
                        return V[j]+int(off[y])+jx
                    if len(ent)==0: z0=z1=10**6
                    elif len(ent)==1: z0=z1=int(ent[0])
                    else: z0,z1=int(ent[0]),int(ent[1])
                    l0.append(look(z0)); l1.append(look(z1))
                    if len(ent)>2:
                        it=max(it,(len(ent)+1)//2)
                        z2=int(ent[2]); z3=int(ent[3]) if len(ent)>3 else z2
                        m0.append(look(z2)); m1.append(look(z3))
                    else: m0.append(None); m1.append(None)
                for lst,key in ((da,'desc'),(db,'desc'),(l0,'look'),(l1,'look')):
                    t,e=conflicts(lst); res[key][0]+=t; res[key][1]+=e
                if it>1:
                    for lst in (m0,m1):
                        t,e=conflicts(lst); res['look2'][0]+=t; res['look2'][1]+=e
                res['slots']+=1; res['iters']+=it
    S=res['slots']
    print(("rowlen(3)=%d "%rowlen(3))+"order=%-5s miss_common=%d desc_inter=%d: iters/slot %.2f | per slot: desc cyc %.2f (extra %.2f) look cyc %.2f (extra %.2f) look2 cyc %.2f (extra %.2f) | total b32 cyc/slot %.2f"%(
        order,miss_common,desc_inter,res['iters']/S,res['desc'][0]/S,res['desc'][1]/S,res['look'][0]/S,res['look'][1]/S,res['look2'][0]/S,res['look2'][1]/S,(res['desc'][0]+res['look'][0]+res['look2'][0])/S))
for order in ("pair",1,8):
    run(order)
run(1,miss_common=True); run(1,desc_inter=True); run(1,True,True); run("pair",True,False); run(8,True,False)

print("-- row lengths padded to odd")
for order in ("pair",8,1): run(order,rowlen=lambda sp:(sp+1)|1)
print("-- row lengths: span+1 rounded so that length mod 4 == 1 or 3, never multiple of 2")
for order in ("pair",8): run(order,rowlen=lambda sp:(sp+1)|1, miss_common=True)

print("-- no guard word (row length = span), misses read one common zero word")
for order in ("pair",8,1): run(order,rowlen=lambda sp:sp, miss_common=True)
