import time, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import quantum_basis_amd as q
from quantum_basis_amd import lattices
import test_gpu_configs as T
t=time.time(); A=T._sector((1,0)); A.sync(); print("gen k10 %.1f" % (time.time()-t)); 
t=time.time(); ia,ja,val=A.download(0,200000); print("download %.1f"%(time.time()-t))
t=time.time(); T._herm_lin(A, complex_x=True); A.sync(); print("herm_lin %.1f"%(time.time()-t))
t=time.time(); maxit=64; v,hess=A.vec(2),np.zeros(2*maxit); A.randomize(v.at(0),1); m=q.lanczos(0,60,maxit,A.dim,A,None,hess,"sr_val0",device_v=v); v.free(); print("60 steps %.1f"%(time.time()-t))
t=time.time(); A.destroy(); print("destroy %.1f"%(time.time()-t))
t=time.time(); B=T._sector((0,0)); B.sync(); print("gen k00 %.1f"%(time.time()-t))
t=time.time(); rb=q.locate_E0_lanczos(B,nev=1,ncv=0,maxit=600); print("lanczos k00 %.1f steps %s"%(time.time()-t, rb.steps)); 
t=time.time(); B.destroy(); print("destroy %.1f"%(time.time()-t))
