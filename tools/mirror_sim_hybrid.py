"""tools/mirror_sim.py's model with a third policy: a crossing constraint c = (a in G, b in H) is mirrored only if neither
neighbour of c on b's chain is executed in G today - where b meets two constraints of G in a row its velocity stays in G as it does
now.  Usage: python tools/mirror_sim_hybrid.py 40|420   (after tools/mirror_sim_dump.py)"""
import sys, numpy as np
which=sys.argv[1]
d=np.load('/root/repo/gpurun_out/mirror_dump_%s.npz'%which)
a,b,x=d['a'].astype(np.int64),d['b'].astype(np.int64),d['x'].astype(np.float64)
n,C=len(x),len(a); nb=1024; B=n//nb
ext=x.max(0)-x.min(0); side=(ext.prod()/B)**(1/3)
fx=max(1,int(round(ext[0]/side))); fy=max(1,int(round(ext[1]/side)))
while B%fx: fx-=1
while (B//fx)%fy: fy-=1
fz=B//fx//fy
blk=np.zeros(n,np.int64)
ox=np.argsort(x[:,0],kind='stable')
for ix,slab in enumerate(np.array_split(ox,fx)):
    oy=slab[np.argsort(x[slab,1],kind='stable')]
    for iy,row in enumerate(np.array_split(oy,fy)):
        oz=row[np.argsort(x[row,2],kind='stable')]
        for iz,r_ in enumerate(np.array_split(oz,fz)): blk[r_]=(ix*fy+iy)*fz+iz
ba=blk[a]; bb=np.where(b>=0,blk[np.maximum(b,0)],-1)
cross=(b>=0)&(ba!=bb)
L,M,S=0.77,4.38,1.40
chains=[[] for _ in range(n)]
for c in range(C):
    chains[a[c]].append(c)
    if b[c]>=0: chains[b[c]].append(c)
# mirror a crossing constraint only if neither neighbour of c on b's chain runs in block(a(c)) today
mir=np.zeros(C,bool)
for body in range(n):
    ch=chains[body]; k=len(ch)
    for t,c in enumerate(ch):
        if b[c]==body and cross[c]:
            p,q=ch[(t-1)%k],ch[(t+1)%k]
            if (k==1) or (ba[p]!=ba[c] and ba[q]!=ba[c]): mir[c]=True
print(which,"crossing %.1f %%, mirrored %.1f %% of crossing"%(cross.mean()*100, mir.sum()/max(cross.sum(),1)*100))
def run(mirror_mask,iters=10):
    A,Bv,BA,BB,MR=a.tolist(),b.tolist(),ba.tolist(),bb.tolist(),mirror_mask.tolist()
    la=[-1]*n; ls=[0]*n; ea=[0.0]*C; eb=[0.0]*C
    for it in range(iters):
        for c in range(C):
            i,j=A[c],Bv[c]
            def pred(body):
                p=la[body]
                if p<0: return 0.0,-2
                if ls[body]==0: return ea[p],BA[p]
                return eb[p],(BB[p] if MR[p] else BA[p])
            ta,ga=pred(i)
            if j>=0: tb,gb=pred(j)
            if j<0 or not MR[c]:
                g=BA[c]
                st=ta+(L if ga in (g,-2) else M)
                if j>=0: st=max(st,tb+(L if gb in (g,-2) else M))
                ea[c]=eb[c]=st+S
            else:
                gA,gB=BA[c],BB[c]
                ea[c]=max(ta+(L if ga in (gA,-2) else M), tb+(L if gb in (gA,-2) else M))+S
                eb[c]=max(tb+(L if gb in (gB,-2) else M), ta+(L if ga in (gB,-2) else M))+S
            la[i]=c; ls[i]=0
            if j>=0: la[j]=c; ls[j]=1
    return max(max(ea),max(eb))
print(which,"today %.1f"%run(np.zeros(C,bool)),"all mirrored %.1f"%run(cross),"hybrid %.1f"%run(mir),flush=True)
