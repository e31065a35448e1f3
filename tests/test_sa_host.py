"""Host side of the algebraic hierarchy (round 6, no GPU): the irregular-partition generator, the aggregation of pmh_mg_create_sa (host routine pmh_sa_aggregate, csrc/mgsa.hip)
against its scipy restatement (permon_amd.feti.sa_aggregate), and the restated hierarchy as the PC of the CPU oracle's block CG (oracle/mg_host.py): K^+ on blocks that
are not boxes in <= 25 iterations where Jacobi needs hundreds -- the reference inverts ANY block (src/mat/impls/inv/matinv.c:481-580, :734-743)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

from permon_amd import _lib, feti


@pytest.mark.parametrize("kind", ["staircase", "lshape"])
def test_irregular_partition_subdomains_float_with_six_modes(kind):
    """Every subdomain of the irregular cuts is face-connected, no box, and its stiffness matrix has exactly the 6 rigid-body modes MeshFeti hands over as R."""
    es = feti.irregular_partition(4, kind)
    f = feti.MeshFeti(es, contact=True)
    assert f.nsub == 8 and f.n_lambda == f.n_eq + f.n_ineq and f.n_ineq > 0
    rs = f.block_rowstart
    sizes = np.diff(rs)
    assert len(set(sizes.tolist())) > 1  # not eight congruent cubes
    for s in range(f.nsub):
        Kb = f.blocks[s].toarray()
        w = np.linalg.eigvalsh(Kb)
        assert int((np.abs(w) < 1e-10 * w.max()).sum()) == 6
        Rb = f.R[:, rs[s]:rs[s + 1]]
        assert np.abs(Kb @ Rb.T).max() < 1e-12 and np.allclose(Rb @ Rb.T, np.eye(6), atol=1e-12)
        X = f.coords[s]
        ext = X.max(axis=0) - X.min(axis=0)
        nbox = int(round(np.prod(ext / f.h + 1)))
        assert nbox > X.shape[0]  # fewer nodes than its bounding box: not a box
    # the gluing couples every copy of a shared dof (rows sum to zero on a rigid translation of the whole body)
    t = np.zeros(f.N)
    t[0::3] = 1.0
    Bt = f.B @ t
    assert np.abs(Bt[f.n_dirichlet:f.n_eq]).max() < 1e-12


def _strength(A, bs, theta):
    Ac = A.tocoo()
    nn = A.shape[0] // bs
    S2 = sp.coo_matrix((Ac.data ** 2, (Ac.row // bs, Ac.col // bs)), shape=(nn, nn)).tocsr()
    S2.sum_duplicates()
    S2.sort_indices()
    dg = np.sqrt(S2.diagonal())
    Sc = S2.tocoo()
    sv = np.sqrt(Sc.data)
    keep = (Sc.row != Sc.col) & (sv > theta * np.sqrt(dg[Sc.row] * dg[Sc.col])) & (sv > 0)
    S = sp.csr_matrix((sv[keep], (Sc.row[keep], Sc.col[keep])), shape=(nn, nn))
    S.sort_indices()
    return S


@pytest.mark.parametrize("theta", [0.0, 0.08])
def test_library_aggregates_equal_the_restatement(theta):
    L = _lib.load()
    f = feti.MeshFeti(feti.irregular_partition(5, "staircase"), contact=False)
    for s in (0, 4, 7):
        A = f.blocks[s].tocsr()
        A.sort_indices()
        n = A.shape[0]
        agg = np.zeros(n // 3, dtype=np.int32)
        na = C.c_int()
        ip, ci = np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32)
        _lib.check(L.pmh_sa_aggregate(n, 3, ip.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), A.data.ctypes.data_as(C.c_void_p), theta, agg.ctypes.data_as(C.c_void_p), C.byref(na)))
        ref, nref = feti.sa_aggregate(_strength(A, 3, theta))
        assert na.value == nref and np.array_equal(agg, ref)
        cnt = np.bincount(agg)
        assert cnt.min() >= 3 and cnt.max() <= 64  # enough nodes to carry six modes, no runaway aggregate


def test_restated_hierarchy_preconditions_the_oracle_block_cg():
    from oracle import mg_host

    f = feti.MeshFeti(feti.irregular_partition(6, "staircase"), contact=False)
    rs = f.block_rowstart
    sel = [1, 6]
    blocks = [f.blocks[s] for s in sel]
    nns = [f.R[:, rs[s]:rs[s + 1]] for s in sel]
    H = feti.sa_mg_hierarchy(blocks, nns, ndof=3, max_coarse=300, theta=0.08)
    assert len(H["A"]) >= 2 and all(a.shape[0] % 3 == 0 for a in H["A"])
    # the coarse operators keep the blocks' kernels: P reproduces the rigid-body modes exactly, so A_c R_c = 0
    for l, P in enumerate(H["P"]):
        assert abs(H["A"][l] - H["A"][l].T).max() < 1e-10
    K = feti.csr_block_diag(blocks)
    brs = np.concatenate([[0], np.cumsum([b.shape[0] for b in blocks])])
    R = np.concatenate(nns, axis=1)
    rhs = np.random.default_rng(2).standard_normal(K.shape[0])
    kp = mg_host.KplusMG(K, brs, H, R=R, rtol=1e-12, max_it=200)
    u = kp(rhs)
    assert kp.last_its <= 25, kp.last_its
    for k, s in enumerate(sel):
        ref = np.linalg.pinv(f.blocks[s].toarray(), rcond=1e-10, hermitian=True) @ rhs[brs[k]:brs[k + 1]]
        assert np.linalg.norm(u[brs[k]:brs[k + 1]] - ref) <= 1e-9 * np.linalg.norm(ref)


@pytest.mark.parametrize("physics,ndof,kind,n", [("elasticity", 3, "staircase", 6), ("elasticity", 3, "lshape", 5), ("poisson", 1, "staircase", 6)])
def test_library_hierarchy_on_the_host_vs_restatement(physics, ndof, kind, n):
    """pmh_sa_hierarchy_host (the whole host builder of pmh_mg_create_sa for one block, no device): the levels have the sizes of the scipy restatement, a floating block stays
    consistently singular down the hierarchy (A_l B_l = 0), every prolongation reproduces the kernel vectors (P_l B_{l+1} = B_l) and the coarsest operator's dense pseudo-inverse
    is one (A A^+ A = A).  Also what scripts/asan_hostlib.sh runs the builder's index arithmetic through."""
    L = _lib.load()
    f = feti.MeshFeti(feti.irregular_partition(n, kind), physics=physics, contact=False)
    rs = f.block_rowstart
    for s in (0, 3, 7):
        A = f.blocks[s].tocsr()
        A.sort_indices()
        nb = A.shape[0]
        Rb = np.ascontiguousarray(f.R[:, rs[s]:rs[s + 1]])
        ip, ci = np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32)
        for maxc in (200, 40):
            nl = C.c_int()
            rows = np.zeros(16, dtype=np.int32)
            defect = np.zeros(3)
            _lib.check(L.pmh_sa_hierarchy_host(nb, ndof, ip.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), A.data.ctypes.data_as(C.c_void_p), Rb.shape[0], Rb.ctypes.data_as(C.c_void_p), maxc, 0.08,
                                               C.byref(nl), rows.ctypes.data_as(C.c_void_p), defect.ctypes.data_as(C.c_void_p)))
            H = feti.sa_mg_hierarchy([A], [Rb], ndof=ndof, max_coarse=maxc, theta=0.08)
            assert rows[:nl.value].tolist() == [a.shape[0] for a in H["A"]], (s, maxc, rows[:nl.value], [a.shape[0] for a in H["A"]])
            assert defect[0] <= 1e-12 and defect[1] <= 1e-12 and defect[2] <= 1e-9, defect
    # a non-singular block (Dirichlet dofs as identity rows): translations as near-kernel, plain inverse at the bottom
    Kb = f.blocks[0].tolil()
    fix = np.arange(0, 3 * ndof)
    keep = np.ones(Kb.shape[0])
    keep[fix] = 0.0
    Kn = (sp.diags(keep) @ f.blocks[0] @ sp.diags(keep) + sp.diags(1.0 - keep)).tocsr()
    Kn.eliminate_zeros()
    Kn.sort_indices()
    ip, ci = np.ascontiguousarray(Kn.indptr, dtype=np.int32), np.ascontiguousarray(Kn.indices, dtype=np.int32)
    nl, rows, defect = C.c_int(), np.zeros(16, dtype=np.int32), np.zeros(3)
    _lib.check(L.pmh_sa_hierarchy_host(Kn.shape[0], ndof, ip.ctypes.data_as(C.c_void_p), ci.ctypes.data_as(C.c_void_p), Kn.data.ctypes.data_as(C.c_void_p), 0, None, 40, 0.08, C.byref(nl),
                                       rows.ctypes.data_as(C.c_void_p), defect.ctypes.data_as(C.c_void_p)))
    assert nl.value >= 2 and rows[0] == Kn.shape[0] and all(rows[l + 1] < rows[l] for l in range(nl.value - 1)) and defect[2] <= 1e-9
