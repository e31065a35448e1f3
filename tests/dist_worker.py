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
    # 128-byte communicator id broadcast (what bench.py does with the ncclUniqueId)
    idt = torch.arange(128, dtype=torch.uint8) if rank == 0 else torch.zeros(128, dtype=torch.uint8)
    dist.broadcast(idt, 0)
    assert bytes(idt.tolist()) == bytes(range(128))
    dist.barrier()
    dist.destroy_process_group()
    print("rank %d ok" % rank)


if __name__ == "__main__":
    main()
