"""GPU tests of the multigrid preconditioner of the MATINV inner CG (pmh_mg_*, pmh_matinv_set_pc_mg):
the V-cycle against a numpy restatement of the same cycle, K^+ with the V-cycle PC against the dense pseudo-inverse and
against the Jacobi-preconditioned path, and a contact TFETI solve that must not notice which PC the inner KSP uses."""
import numpy as np
import pytest

import permon_amd as pa
from permon_amd.chain import FetiDualQP
from permon_amd.feti import CubeFeti, DmdaFeti, box_mg_hierarchy

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = pa.Context(0)
    yield c
    c.close()


def _vcycle_numpy(H, degree, lo, hi):
    from oracle.mg_host import vcycle  # the CPU restatement of the cycle (test infrastructure)

    return vcycle(H, degree, lo, hi)


def _cube_hierarchy(f, min_nodes=27):
    """min_nodes=27 forces the deepest hierarchy on these small cubes (the default stops at <= 400 nodes per block)."""
    nn = f.nel + 1
    return box_mg_hierarchy([f.Ki] * f.nsub, [(nn, nn, nn)] * f.nsub, f.ndof, min_nodes=min_nodes)


@pytest.mark.parametrize("physics,degree,csr_only", [("poisson", 2, False), ("elasticity", 2, False), ("elasticity", 3, False), ("elasticity", 2, True), ("elasticity", 2, "unfused")])
def test_vcycle_matches_numpy_restatement(ctx, physics, degree, csr_only, monkeypatch):
    """fp64 cycle: the elasticity levels run on the 3x3-block kernel (bsr.hip) unless csr_only, Poisson on the CSR kernel.
    Degree 2 on block operators finishes the smoothing steps inside the operator kernel ("unfused" switches that off)."""
    if csr_only == "unfused":
        monkeypatch.setenv("PMH_MG_FUSED", "0")
        csr_only = False
    f = CubeFeti((2, 1, 1), 8, physics, contact=False)
    H = _cube_hierarchy(f)
    assert len(H["A"]) >= 3
    if csr_only:
        monkeypatch.setenv("PMH_MG_NO_BSR", "1")
    mg = pa.MG(ctx, H, degree=degree)
    b = np.random.default_rng(5).standard_normal(f.N)
    x = ctx.vec(f.N)
    mg.apply(ctx.vec_from(b), x)
    ref = _vcycle_numpy(H, degree, 0.1, 1.1)(b)
    assert np.linalg.norm(x.to_numpy() - ref) <= 1e-11 * np.linalg.norm(ref)
    # symmetric: <V a, b> == <a, V b>
    a = np.random.default_rng(6).standard_normal(f.N)
    y = ctx.vec(f.N)
    mg.apply(ctx.vec_from(a), y)
    assert abs(y.to_numpy() @ b - a @ x.to_numpy()) <= 1e-10 * abs(a @ x.to_numpy())


def test_fp32_cycle_is_a_close_copy_of_the_fp64_cycle(ctx):
    f = CubeFeti((2, 1, 1), 8, "elasticity", contact=False)
    H = _cube_hierarchy(f)
    b = np.random.default_rng(7).standard_normal(f.N)
    out = []
    for prec in ("fp64", "fp32", "fp16"):
        mg = pa.MG(ctx, H, precision=prec)
        x = ctx.vec(f.N)
        mg.apply(ctx.vec_from(b), x)
        out.append(x.to_numpy())
    assert np.linalg.norm(out[1] - out[0]) <= 2e-5 * np.linalg.norm(out[0])
    assert np.linalg.norm(out[2] - out[0]) <= 5e-3 * np.linalg.norm(out[0])  # fp16 entries: 2^-11 relative perturbation of K
    with pytest.raises(pa.PermonHipError):  # Poisson blocks have no 3x3 structure
        g = CubeFeti((2, 1, 1), 8, "poisson", contact=False)
        pa.MG(ctx, _cube_hierarchy(g), precision="fp32")


def test_matinv_bsr3_product_matches_csr(ctx):
    f = CubeFeti((2, 1, 1), 7, "elasticity", contact=False)
    K = pa.MatBlockDiag.from_scipy(ctx, f.block_rowstart, f.K)
    rhs = np.random.default_rng(4).standard_normal(f.N)
    u0, u1 = ctx.vec(f.N), ctx.vec(f.N)
    M0 = pa.MatInv(K, rtol=1e-12, nullspace=f.R)
    M0.mult(ctx.vec_from(rhs), u0)
    M1 = pa.MatInv(K, rtol=1e-12, nullspace=f.R)
    M1.enable_bsr3()
    M1.timing_enable(4000)
    M1.mult(ctx.vec_from(rhs), u1)
    assert abs(M1.last_iterations()[0] - M0.last_iterations()[0]) <= 2
    assert np.linalg.norm(u1.to_numpy() - u0.to_numpy()) <= 1e-10 * np.linalg.norm(u0.to_numpy())
    n, ms, nbytes = M1.timing_get()
    assert n >= M1.last_iterations()[0] and ms > 0
    # HBM bytes of a launch: the two cubes are congruent and share ONE device copy (streamed once, 8.44 B per non-zero) + x + y of both
    nrep = M1.bsr3_replicas()
    assert nrep == 2
    assert abs(nbytes - (f.K.nnz // 9 // nrep * 76 + 4 * (f.N // 3 // nrep + 1) + 16 * f.N)) <= 76 * 64
    # a Poisson K has no 3x3 blocks: refused loudly
    g = CubeFeti((2, 1, 1), 4, "poisson", contact=False)
    Kg = pa.MatBlockDiag.from_scipy(ctx, g.block_rowstart, g.K)
    with pytest.raises(pa.PermonHipError):
        pa.MatInv(Kg).enable_bsr3()


@pytest.mark.parametrize("nel,prec", [(8, "fp64"), (11, "fp64"), (8, "fp32"), (11, "fp32"), (11, "fp16")])  # 11: odd element count, non-nested last coarse interval
def test_matinv_with_mg_pc(ctx, nel, prec):
    f = CubeFeti((2, 1, 1), nel, "elasticity", contact=False)
    K = pa.MatBlockDiag.from_scipy(ctx, f.block_rowstart, f.K)
    rhs = np.random.default_rng(1).standard_normal(f.N)
    uj, um = ctx.vec(f.N), ctx.vec(f.N)
    Mj = pa.MatInv(K, rtol=1e-12, nullspace=f.R)
    Mj.mult(ctx.vec_from(rhs), uj)
    its_j, _ = Mj.last_iterations()
    Mm = pa.MatInv(K, rtol=1e-12, nullspace=f.R)
    if prec != "fp64":
        Mm.enable_bsr3()
    mg = Mm.set_pc_mg(_cube_hierarchy(f), precision=prec)
    Mm.mult(ctx.vec_from(rhs), um)
    its_m, spmv_m = Mm.last_iterations()
    assert its_m <= 20 and its_m * 4 < its_j  # mesh-independent and far below the Jacobi count
    assert mg.fine_spmv() >= 4 * (its_m + 1) and spmv_m >= its_m  # 2*degree fine SpMVs per V-cycle, one V-cycle per CG step + start
    ref = uj.to_numpy()
    assert np.linalg.norm(um.to_numpy() - ref) <= 1e-9 * np.linalg.norm(ref)
    # Moore-Penrose: K u = P_R rhs and R'u = 0
    u = um.to_numpy()
    Rm = f.kernel_matrix()
    rp = rhs - Rm @ (Rm.T @ rhs)
    assert np.linalg.norm(f.K @ u - rp) <= 1e-9 * np.linalg.norm(rp)
    assert np.linalg.norm(Rm.T @ u) <= 1e-9 * np.linalg.norm(u)
    # second application reuses the hierarchy and the iteration-count hint
    Mm.mult(ctx.vec_from(2.0 * rhs), um)
    assert np.linalg.norm(um.to_numpy() - 2.0 * ref) <= 1e-9 * np.linalg.norm(ref) * 2.0


def test_mg_on_heterogeneous_dmda_blocks(ctx):
    """ex71 elasticity slabs: blocks of different sizes, one of them non-singular (Dirichlet in the matrix), thin in x."""
    prob = DmdaFeti((8, 6, 4), 7, "elasticity")
    dims = [(2, 7, 5), (3, 7, 5)] + [(2, 7, 5)] * 5
    assert [d[0] * d[1] * d[2] * 3 for d in dims] == [K.shape[0] for K in prob.blocks]
    H = box_mg_hierarchy(prob.blocks, dims, 3, min_nodes=27)
    K = pa.MatBlockDiag.from_scipy(ctx, prob.block_rowstart, prob.K)
    rhs = np.random.default_rng(2).standard_normal(prob.N)
    uj, um = ctx.vec(prob.N), ctx.vec(prob.N)
    Mj = pa.MatInv(K, rtol=1e-12, nullspace=prob.R)
    Mj.mult(ctx.vec_from(rhs), uj)
    Mm = pa.MatInv(K, rtol=1e-12, nullspace=prob.R)
    Mm.set_pc_mg(H)
    Mm.mult(ctx.vec_from(rhs), um)
    assert Mm.last_iterations()[0] < Mj.last_iterations()[0]
    assert np.linalg.norm(um.to_numpy() - uj.to_numpy()) <= 1e-8 * np.linalg.norm(uj.to_numpy())


def test_contact_tfeti_solve_is_independent_of_the_inner_pc(ctx):
    f = CubeFeti((2, 2, 1), 6, "elasticity", contact=True)
    G, e = f.coarse()
    sols, its = [], []
    for use_mg in (False, True):
        q = FetiDualQP(ctx, f.subset(range(f.nsub)), G, e, f.c, f.lb, orthonormal=True, kplus_rtol=1e-11)
        if use_mg:  # the production configuration: fp32 V-cycle + block-kernel K x in the CG
            q.Kplus.enable_bsr3()
            q.Kplus.set_pc_mg(_cube_hierarchy(f, min_nodes=400), precision="fp32")  # default depth: 2 levels here
        st = q.solve_smalxe(rtol=1e-6)
        assert st.reason > 0
        sols.append(q.dual_solution())
        its.append((st.iteration, st.inner_iter_accu))
    assert its[0] == its[1]  # same outer / inner MPGP counts: K^+ is the same operator to 1e-11
    assert np.linalg.norm(sols[0] - sols[1]) <= 1e-7 * np.linalg.norm(sols[0])


@pytest.mark.parametrize("case", ["floating", "regularized", "odd_box"])
def test_box_hierarchy_built_in_c(ctx, case):
    """pmh_mg_create_box (host C++ inside libpermonhip: trilinear prolongation (x) I3, Galerkin products, coarse pseudo-inverses from the
    injected kernel) against the scipy builder of permon_amd.feti: same K^+ to 1e-9 and the same CG iteration count (+-2)."""
    if case == "odd_box":
        f = CubeFeti((2, 1, 1), 9, "elasticity", contact=False)  # 10 nodes per edge: short last coarse interval
    else:
        f = CubeFeti((2, 1, 1), 8, "elasticity", contact=False)
    nn = f.nel + 1
    local = f.subset(range(f.nsub))
    if case == "regularized":
        from permon_amd.chain import regularize_blocks

        Ksp = regularize_blocks(ctx, local)[0]
        R = None
    else:
        Ksp, R = f.K, f.R
    blocks = [Ksp[i * f.n_i:(i + 1) * f.n_i, i * f.n_i:(i + 1) * f.n_i].tocsr() for i in range(f.nsub)]
    H = box_mg_hierarchy(blocks, [(nn, nn, nn)] * f.nsub, 3, min_nodes=27)
    rhs = np.random.default_rng(8).standard_normal(f.N)
    out = []
    for builder in ("python", "c"):
        K = pa.MatBlockDiag.from_scipy(ctx, f.block_rowstart, Ksp)
        M = pa.MatInv(K, rtol=1e-11, nullspace=R)
        if builder == "python":
            M.set_pc_mg(H, degree=2)
        else:
            M.set_pc_mg_box(Ksp, [(nn, nn, nn)] * f.nsub, 3, R=R, min_nodes=27, degree=2)
        u = ctx.vec(f.N)
        M.mult(ctx.vec_from(rhs), u)
        out.append((u.to_numpy(), M.last_iterations()[0]))
    (ua, ia), (ub, ib) = out
    assert np.linalg.norm(ua - ub) <= 1e-9 * np.linalg.norm(ua) and abs(ia - ib) <= 2 and ib < 30
    # and against the dense (pseudo-)inverse
    Kd = blocks[0].toarray()
    ref1 = np.linalg.pinv(Kd, rcond=1e-10, hermitian=True) if case != "regularized" else None
    if ref1 is not None:
        ref = np.concatenate([ref1 @ rhs[:f.n_i], ref1 @ rhs[f.n_i:]])
        assert np.linalg.norm(ub - ref) <= 1e-8 * np.linalg.norm(ref)
    else:
        assert np.linalg.norm(Ksp @ ub - rhs) <= 1e-9 * np.linalg.norm(rhs)
