"""World-size-2 worker (gloo, CPU): checks the sharding plan of the FETI path that bench.py / the RCCL build use.

Each rank takes its contiguous share of the subdomain blocks (CubeFeti.subset), applies ITS part of
F = B K^+ B' with the CPU oracle (local B' lambda, local block-wise K^+, local B u) and the partial B u are summed with
one all-reduce over the replicated lambda -- exactly the one collective of the GPU path (pmh_gluing_mult_transpose).
The sum must equal the single-rank operator; replicated dual-space results must be identical on both ranks."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from permon_amd.feti import CubeFeti  # noqa: E402


def main():
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % os.environ["MASTER_PORT"], rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    f = CubeFeti((2, 2, 1), 2, contact=True)
    per = f.nsub // world
    assert per * world == f.nsub
    loc = f.subset(range(rank * per, (rank + 1) * per))
    # every local leaf points into this rank's primal range, dual numbering stays global
    assert loc["leaves_row"].min() >= 0 and loc["leaves_row"].max() < loc["n_x"]
    assert loc["n_lambda"] == f.n_lambda
    nleaf = torch.tensor([len(loc["leaves_row"])])
    dist.all_reduce(nleaf)
    assert int(nleaf) == len(f.leaves_row)  # the leaves are partitioned, none lost or duplicated

    K = O.Csr.from_scipy(loc["K"])
    Kp = O.MatInv(K, loc["block_rowstart"], loc["R"], rtol=1e-13)
    B = O.Gluing(loc["n_x"], f.n_lambda, loc["leaves_row"], loc["leaves_root"], loc["leaves_sign"])
    lam = np.random.default_rng(7).standard_normal(f.n_lambda)  # replicated: same seed on every rank
    part = B.mult_transpose(Kp.mult(B.mult(lam)))
    t = torch.from_numpy(part.copy())
    dist.all_reduce(t)  # the single data-path collective
    y = t.numpy()

    if rank == 0:
        Kg = O.Csr.from_scipy(f.K)
        Kpg = O.MatInv(Kg, f.block_rowstart, f.R, rtol=1e-13)
        Bg = O.Gluing(f.N, f.n_lambda, f.leaves_row, f.leaves_root, f.leaves_sign)
        ref = Bg.mult_transpose(Kpg.mult(Bg.mult(lam)))
        assert np.linalg.norm(y - ref) <= 1e-11 * np.linalg.norm(ref), np.linalg.norm(y - ref)
    # replicated results are bit-identical across ranks (all-reduce returns the same bits everywhere)
    chk = torch.tensor([float(np.sum(y)), float(np.dot(y, y))], dtype=torch.float64)
    lst = [torch.zeros_like(chk) for _ in range(world)]
    dist.all_gather(lst, chk)
    assert all(torch.equal(lst[0], v) for v in lst)
    # ---- the striped explicit operators (pmh_fexplicit_set_stripe): every rank applies the 128-row stripes the library's dealing rule gives it
    # -- of ALL blocks -- and the same all-reduce completes F lambda.  The plan comes from libpermonhip's host helper (no GPU needed);
    # W_b = pinv(K_b)[Gamma_b, Gamma_b] by numpy stands in for the assembled blocks.
    import ctypes as C

    import permon_amd as pa

    L = pa.load()
    g = CubeFeti((2, 1, 1), 7, contact=True)  # n_Gamma > 128: several stripes per block
    Kp = np.linalg.pinv(g.Ki.toarray(), rcond=1e-10, hermitian=True)
    Bd = g.B.toarray()
    gam = [np.nonzero(np.abs(Bd[:, s * g.n_i:(s + 1) * g.n_i]).sum(axis=0))[0] for s in range(g.nsub)]
    ng = np.array([len(x) for x in gam], dtype=np.int32)
    assert ng.min() > 128
    nst = [int(-(-n // 128)) for n in ng]
    owner = np.zeros(sum(nst), dtype=np.int32)
    pa._lib.check(L.pmh_fexplicit_stripe_owner(g.nsub, ng.ctypes.data_as(C.c_void_p), world, owner.ctypes.data_as(C.c_void_p)))
    assert set(owner.tolist()) == set(range(world))  # every rank gets stripes
    lam2 = np.random.default_rng(9).standard_normal(g.n_lambda)
    part2, o = np.zeros(g.n_lambda), 0
    for s in range(g.nsub):
        Bs = Bd[:, s * g.n_i:(s + 1) * g.n_i][:, gam[s]]  # Bhat_s
        W = Kp[np.ix_(gam[s], gam[s])]
        xh = Bs.T @ lam2
        yh = np.zeros(ng[s])
        for k in range(nst[s]):
            if owner[o + k] == rank:
                yh[128 * k:128 * (k + 1)] = W[128 * k:128 * (k + 1)] @ xh  # the rows of this stripe
        o += nst[s]
        part2 += Bs @ yh
    t2 = torch.from_numpy(part2.copy())
    dist.all_reduce(t2)
    if rank == 0:
        Fd = sum(Bd[:, s * g.n_i:(s + 1) * g.n_i] @ Kp @ Bd[:, s * g.n_i:(s + 1) * g.n_i].T for s in range(g.nsub))
        assert np.linalg.norm(t2.numpy() - Fd @ lam2) <= 1e-11 * np.linalg.norm(Fd @ lam2)
    # ---- the class-shared symmetric storage (PMH_FX_CLASS_SYM): ONE W_c for the congruent cubes; a rank owns whole mega bands of 1024 rows of
    # its lower block-triangle and applies both products of every stored tile (rows of its band: direct; the columns they touch: transposed)
    nc = 2600  # 3 mega bands
    nmb = -(-(-(-nc // 256)) // 4)
    own = np.zeros(nmb, dtype=np.int32)
    pa._lib.check(L.pmh_fexplicit_class_sym_plan(nc, world, own.ctypes.data_as(C.c_void_p), None))
    assert set(own.tolist()) == set(range(world))
    rngc = np.random.default_rng(4)
    Wc = rngc.standard_normal((nc, nc))
    Wc = Wc + Wc.T
    Xc = rngc.standard_normal((nc, 8))
    Lw = np.tril(Wc, -1) + 0.5 * np.diag(np.diag(Wc))  # what the tiles hold: strict lower triangle + half the diagonal
    Yc = np.zeros((nc, 8))
    for m in range(nmb):
        if own[m] == rank:
            R = slice(1024 * m, min(nc, 1024 * (m + 1)))
            Yc[R] += Lw[R] @ Xc  # direct sums of the band's rows
            Yc += Lw[R].T @ Xc[R]  # transposed sums on the columns of the band's tiles
    tc = torch.from_numpy(Yc.copy())
    dist.all_reduce(tc)
    assert np.linalg.norm(tc.numpy() - Wc @ Xc) <= 1e-12 * np.linalg.norm(Wc @ Xc)
    # ---- the orbit storage (PMH_FX_CLASS_ORBIT): every rank keeps all representatives' rows and multiplies a contiguous share of the k range (chunks
    # of 16 columns, [nkc r / N, nkc (r + 1) / N)); the partial Y need no collective of their own -- the all-reduce that ends B Y sums them
    Mr, Kc_, Nc = 37, 1000, 24
    Af, Bf = rngc.standard_normal((Mr, Kc_)), rngc.standard_normal((Kc_, Nc))
    nkc = -(-Kc_ // 16)
    k0, k1 = 16 * (nkc * rank // world), min(Kc_, 16 * (nkc * (rank + 1) // world))
    tp = torch.from_numpy(Af[:, k0:k1] @ Bf[k0:k1])
    dist.all_reduce(tp)
    assert np.linalg.norm(tp.numpy() - Af @ Bf) <= 1e-12 * np.linalg.norm(Af @ Bf)
    # 128-byte communicator id broadcast (what bench.py does with the ncclUniqueId)
    idt = torch.arange(128, dtype=torch.uint8) if rank == 0 else torch.zeros(128, dtype=torch.uint8)
    dist.broadcast(idt, 0)
    assert bytes(idt.tolist()) == bytes(range(128))
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
