import sys, random, itertools
sys.path.insert(0,'.')
from quantum_basis_amd import lattices
def stats(bonds, n=36):
    cut=sum(1 for a,b in bonds if (a<18)!=(b<18))
    hh=sum(1 for a,b in bonds if a>=18 and b>=18); ll=sum(1 for a,b in bonds if a<18 and b<18)
    # tiers by max site
    tiers=[0]*4
    for a,b in bonds:
        m=max(a,b); tiers[0 if m<18 else 1 if m<24 else 2 if m<30 else 3]+=1
    mixed_low=sum(1 for a,b in bonds if max(a,b)>=18 and min(a,b)<12)
    return dict(cut=cut,ll=ll,hh=hh,tiers=tiers,mixed_low=mixed_low)
for name,b in (('kagome36a',lattices.kagome_torus((4,2),(2,4))),('kagome36',lattices.kagome(4,3)),('triangular36',lattices.triangular(6,6))):
    print(name,len(b),stats(b))
# optimise labeling for kagome36a: cost = sum over bonds f(min,max)
def cost(perm,bonds):
    c=0.0
    for a,b in bonds:
        x,y=perm[a],perm[b]
        lo,hi=min(x,y),max(x,y)
        if hi<18: continue
        c+= 1.0 if lo>=18 else (2.0 if lo>=12 else 4.0)   # both high: coalesced; mixed: scattered, worse the lower the partner
        if hi>=26 and lo<18: c+=1.0
    return c
def anneal(bonds,n=36,iters=400000,seed=1):
    rnd=random.Random(seed); perm=list(range(n)); cur=cost(perm,bonds); best=(cur,perm[:]); T=2.0
    for it in range(iters):
        i,j=rnd.randrange(n),rnd.randrange(n)
        if i==j: continue
        perm[i],perm[j]=perm[j],perm[i]
        c=cost(perm,bonds)
        if c<=cur or rnd.random()< pow(2.718281828,-(c-cur)/T): 
            cur=c
            if c<best[0]: best=(c,perm[:])
        else: perm[i],perm[j]=perm[j],perm[i]
        T=max(0.02,T*0.99999)
    return best
b=lattices.kagome_torus((4,2),(2,4))
c,perm=anneal(b)
nb=[(perm[x],perm[y]) for x,y in b]
print('optimised',c,stats(nb)); print('perm',perm)
