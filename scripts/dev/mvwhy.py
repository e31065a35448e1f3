import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, time, sys
import permon_amd as pa
from permon_amd import feti
from permon_amd.chain import FetiDualQP
ctx=pa.Context(0)
n=int(sys.argv[1])
f=feti.MeshFeti(feti.irregular_partition(n,"staircase"),contact=True)
G,e=f.coarse(orthonormalize=True)
loc=f.subset(range(8))
t=time.time()
q=FetiDualQP(ctx,loc,G,e,f.c,f.lb,kplus_rtol=1e-9,mg_sa=dict(ndof=3,max_coarse=1500),mg_precision="fp16",bsr3=True)
print("setup %.2f"%(time.time()-t))
F=ctx.vec_from(np.random.default_rng(0).standard_normal(8*f.N)); U=ctx.vec(8*f.N)
try:
    its=q.Kplus.mult_multi(F,U)
    print("multi its",its)
except Exception as ex:
    print("multi failed",ex)
