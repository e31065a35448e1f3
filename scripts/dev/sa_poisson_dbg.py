import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import permon_amd as pa
from permon_amd import feti
from oracle import mg_host
ctx=pa.Context(0)
f=feti.MeshFeti(feti.irregular_partition(6,"staircase"),physics="poisson",contact=True)
rs=f.block_rowstart
blocks=[f.blocks[s] for s in range(8)]
nns=[f.R[:,rs[s]:rs[s+1]] for s in range(8)]
for maxc in (60, 700):
    H=feti.sa_mg_hierarchy(blocks,nns,ndof=1,max_coarse=maxc,theta=0.08)
    print("levels",[a.shape[0] for a in H["A"]])
    K=feti.csr_block_diag(blocks)
    Kd=pa.MatBlockDiag.from_scipy(ctx,rs,K)
    Mi=pa.MatInv(Kd,rtol=1e-12,nullspace=f.R)
    mg=Mi.set_pc_mg_sa(K,1,R=f.R,max_coarse=maxc,precision="fp64")
    V=mg_host.vcycle(H,2)
    b=np.random.default_rng(3).standard_normal(f.N)
    x=ctx.vec(f.N); mg.apply(ctx.vec_from(b),x)
    ref=V(b)
    print("maxc",maxc,"cycle rel diff",np.linalg.norm(x.to_numpy()-ref)/np.linalg.norm(ref))
    # python-built hierarchy through pmh_mg_create
    Mi2=pa.MatInv(Kd,rtol=1e-12,nullspace=f.R)
    mg2=Mi2.set_pc_mg(H,precision="fp64")
    x2=ctx.vec(f.N); mg2.apply(ctx.vec_from(b),x2)
    print("   python hierarchy through pmh_mg_create: rel diff",np.linalg.norm(x2.to_numpy()-ref)/np.linalg.norm(ref))
    u=ctx.vec(f.N); Mi2.mult(ctx.vec_from(b),u); print("   its with python hierarchy",Mi2.last_iterations()[0])
    Mi.mult(ctx.vec_from(b),u); print("   its with C++ hierarchy",Mi.last_iterations()[0])
