import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import permon_amd as pa
from permon_amd import feti
from oracle import mg_host
ctx=pa.Context(0)
f=feti.MeshFeti(feti.irregular_partition(6,"staircase"),physics="poisson",contact=False)
rs=f.block_rowstart
for sel in ([0,1,2,3,4,5,6,7],[0,1,2,3],[4,5,6,7],[0,1],[2,3],[6,7],[5],[6],[7]):
    blocks=[f.blocks[s] for s in sel]
    nns=[f.R[:,rs[s]:rs[s+1]] for s in sel]
    K=feti.csr_block_diag(blocks); brs=np.concatenate([[0],np.cumsum([b.shape[0] for b in blocks])]).astype(np.int32)
    R=np.concatenate(nns,axis=1)
    for maxc in (200,60):
        H=feti.sa_mg_hierarchy(blocks,nns,ndof=1,max_coarse=maxc,theta=0.08)
        Kd=pa.MatBlockDiag.from_scipy(ctx,brs,K)
        Mi=pa.MatInv(Kd,rtol=1e-12,nullspace=R)
        mg=Mi.set_pc_mg_sa(K,1,R=R,max_coarse=maxc,precision="fp64")
        V=mg_host.vcycle(H,2)
        b=np.random.default_rng(3).standard_normal(K.shape[0])
        x=ctx.vec(K.shape[0]); mg.apply(ctx.vec_from(b),x)
        ref=V(b)
        print(sel,"levels",[a.shape[0] for a in H["A"]],"coarse sizes",np.diff(H["coarse_rowstart"]).tolist(),"cycle rel diff %.2e"%(np.linalg.norm(x.to_numpy()-ref)/np.linalg.norm(ref)), flush=True)
