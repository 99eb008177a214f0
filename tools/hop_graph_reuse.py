import numpy as np, itertools, sys
sys.path.insert(0,'.')
from quantum_basis_amd import lattices
bonds=np.asarray(lattices.square(4,4)).reshape(-1,2)
n=16;k=8
def patterns(n,k):
    return np.array(sorted(sum(1<<s for s in c) for c in itertools.combinations(range(n),k)),dtype=np.int64)
P=patterns(n,k); N=len(P); idx={int(p):i for i,p in enumerate(P)}
nb=[[] for _ in range(N)]
for i,p in enumerate(P):
    p=int(p)
    for a,b in bonds:
        a=int(a);b=int(b)
        if ((p>>a)&1)!=((p>>b)&1):
            q=p^((1<<a)|(1<<b)); nb[i].append(idx[q])
nb=[sorted(set(x)) for x in nb]
deg=np.mean([len(x) for x in nb]); print('N',N,'avg deg',deg)
def score(order,B):
    # order: list of node ids; blocks of B consecutive; mean |N(U) u U| / B
    tot=0;cnt=0
    for s in range(0,N,B):
        U=order[s:s+B]
        S=set(U)
        for u in U: S.update(nb[u])
        tot+=len(S);cnt+=len(U)
    return tot/cnt
colex=list(range(N))
print('colex', [round(score(colex,B),2) for B in (1,4,8,16,32,64)])
# order by fixing occupancy of a site subset: sort key = (pattern restricted to subset A, rest)
def key_order(siteperm):
    # relabel sites: new bit position i <- old site siteperm[i]; sort by relabeled value
    vals=[]
    for p in P:
        p=int(p);v=0
        for i,s in enumerate(siteperm): v|=((p>>s)&1)<<i
        vals.append(v)
    return list(np.argsort(vals,kind='stable'))
# snake / block site orders: low bits = one 2x2 plaquette etc.
site=lambda x,y:x+4*y
perm_rows=[site(x,y) for y in range(4) for x in range(4)]
perm_plaq=[site(x+2*bx,y+2*by) for by in range(2) for bx in range(2) for y in range(2) for x in range(2)]
for name,perm in (('rows',perm_rows),('plaq',perm_plaq),('plaq_rev',perm_plaq[::-1])):
    o=key_order(perm); print(name,[round(score(o,B),2) for B in (1,4,8,16,32,64)])
# Gray-code-like: BFS order
from collections import deque
seen=[False]*N;o=[]
dq=deque([0]);seen[0]=True
while dq:
    u=dq.popleft();o.append(u)
    for v in nb[u]:
        if not seen[v]: seen[v]=True;dq.append(v)
print('bfs',[round(score(o,B),2) for B in (1,4,8,16,32,64)])
# greedy clustering: grow block by adding the node that adds fewest new neighbours
import random
def greedy(B):
    left=set(range(N));order=[]
    while left:
        u=min(left);U=[u];left.discard(u);S=set(nb[u])|{u}
        while len(U)<B and left:
            cand=[v for v in S if v in left]
            if not cand: cand=[min(left)]
            best=None;bs=None
            for v in cand[:200]:
                add=len(set(nb[v])-S)
                if bs is None or add<bs: bs=add;best=v
            U.append(best);left.discard(best);S|=set(nb[best]);S.add(best)
        order+=U
    return order
for B in (8,16,32):
    o=greedy(B); print('greedy',B,round(score(o,B),2))
