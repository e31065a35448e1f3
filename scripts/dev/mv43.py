import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, time, ctypes as C
import permon_amd as pa
from permon_amd import feti
from permon_amd.chain import FetiDualQP
from permon_amd._lib import check
ctx=pa.Context(0)
n=int(sys.argv[1]); maxc=int(sys.argv[2]) if len(sys.argv)>2 else 1500
t=time.time()
f=feti.MeshFeti(feti.irregular_partition(n,"staircase"),contact=True)
print("generate %.1f"%(time.time()-t), flush=True)
G,e=f.coarse(orthonormalize=True)
loc=f.subset(range(8))
t=time.time()
q=FetiDualQP(ctx,loc,G,e,f.c,f.lb,kplus_rtol=1e-12,mg_sa=dict(ndof=3,max_coarse=maxc),mg_precision="fp16",bsr3=True)
ctx.sync(); print("setup %.2f"%(time.time()-t), flush=True)
rhs=ctx.vec_from(np.random.default_rng(1).standard_normal(f.N)); u=ctx.vec(f.N)
q.Kplus.mult(rhs,u); ctx.sync(); t=time.time()
for _ in range(3): q.Kplus.mult(rhs,u)
ctx.sync(); dt=(time.time()-t)/3
it1=q.Kplus.last_iterations()[0]
print("single its",it1,"ms per application %.2f"%(dt*1e3),"ms per iteration %.3f"%(dt*1e3/it1), flush=True)
F=ctx.vec_from(np.random.default_rng(0).standard_normal(8*f.N)); U=ctx.vec(8*f.N)
t=time.time(); its=q.Kplus.mult_multi(F,U); ctx.sync(); dt=time.time()-t
print("multi its",its,"ms incl. set-up of the mv solver %.1f"%(dt*1e3), flush=True)
