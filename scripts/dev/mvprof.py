import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, time
import permon_amd as pa
from permon_amd import feti
from permon_amd.chain import FetiDualQP
ctx=pa.Context(0)
n=int(sys.argv[1]); reps=int(sys.argv[2]) if len(sys.argv)>2 else 3
maxc=int(sys.argv[3]) if len(sys.argv)>3 else 1500
f=feti.MeshFeti(feti.irregular_partition(n,"staircase"),contact=True)
G,e=f.coarse(orthonormalize=True)
loc=f.subset(range(8))
t=time.time()
q=FetiDualQP(ctx,loc,G,e,f.c,f.lb,kplus_rtol=1e-12,mg_sa=dict(ndof=3,max_coarse=maxc),mg_precision="fp16",bsr3=True)
ctx.sync(); print("setup %.2f"%(time.time()-t), flush=True)
F=ctx.vec_from(np.random.default_rng(0).standard_normal(8*f.N)); U=ctx.vec(8*f.N)
its=q.Kplus.mult_multi(F,U)
ctx.sync(); t=time.time()
for _ in range(reps): its=q.Kplus.mult_multi(F,U)
ctx.sync(); dt=(time.time()-t)/reps
print("multi its",its,"ms per application %.2f"%(dt*1e3),"ms per iteration %.3f"%(dt*1e3/its), flush=True)
rhs=ctx.vec_from(np.random.default_rng(1).standard_normal(f.N)); u=ctx.vec(f.N)
q.Kplus.mult(rhs,u); ctx.sync(); t=time.time()
for _ in range(reps): q.Kplus.mult(rhs,u)
ctx.sync(); dt=(time.time()-t)/reps
it1=q.Kplus.last_iterations()[0]
print("single its",it1,"ms per application %.2f"%(dt*1e3),"ms per iteration %.3f"%(dt*1e3/it1), flush=True)
