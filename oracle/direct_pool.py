"""CPU baseline infrastructure (TEST INFRASTRUCTURE: only bench.py's cpu_baseline legs may import this; nothing under permon_amd/ does).

The reference's DEFAULT K^+ on the host cores: MATINV factors every subdomain block once (PCCHOLESKY / MUMPS, src/mat/impls/inv/matinv.c:481-580) and
MatMult_Inv (matinv.c:734-743) is one forward / backward substitution per block, one block per MPI rank.  PETSc / MUMPS are absent from the image, so SuperLU
(scipy.sparse.linalg.splu) stands in, and since scipy's solve holds the interpreter lock the blocks are dealt to WORKER PROCESSES -- the ranks of the reference's
run: every worker factors the (congruent) block matrix itself and then solves its share of the blocks' right-hand sides per application, vectors through shared
memory.  The pool is started BEFORE the parent initialises the GPU (spawned interpreters that import numpy / scipy only) and idles until the CPU leg runs.
"""
import multiprocessing as mp
from multiprocessing import shared_memory

import numpy as np


def _worker(conn):
    import scipy.sparse as sp
    import scipy.sparse.linalg as spla

    lu = None
    shm_in = shm_out = None
    X = Y = None
    while True:
        msg = conn.recv()
        if msg[0] == "stop":
            break
        try:
            if msg[0] == "factor":
                _, n, indptr, indices, data = msg
                K = sp.csc_matrix((data, indices, indptr), shape=(n, n))
                lu = spla.splu(K, permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0, options=dict(SymmetricMode=True))
                conn.send(("ok", int(lu.L.nnz + lu.U.nnz)))
            elif msg[0] == "attach":
                _, name_in, name_out, n, ncol = msg
                shm_in, shm_out = shared_memory.SharedMemory(name=name_in), shared_memory.SharedMemory(name=name_out)
                X = np.ndarray((ncol, n), dtype=np.float64, buffer=shm_in.buf)
                Y = np.ndarray((ncol, n), dtype=np.float64, buffer=shm_out.buf)
                conn.send(("ok", 0))
            elif msg[0] == "solve":
                _, c0, c1 = msg
                for a in range(c0, c1, 8):  # (8 right-hand sides per call: what SuperLU's solve does best here)
                    b = min(c1, a + 8)
                    Y[a:b, :] = lu.solve(np.ascontiguousarray(X[a:b, :].T)).T
                conn.send(("ok", c1 - c0))
        except Exception as ex:  # noqa: BLE001
            conn.send(("error", repr(ex)))
    for s in (shm_in, shm_out):
        if s is not None:
            s.close()


class DirectPool:
    """nworkers processes, each holding its own SuperLU factors of ONE block matrix; solve(F) = K^{-1} F for `ncol` right-hand sides dealt evenly."""

    def __init__(self, nworkers):
        ctx = mp.get_context("spawn")
        self.nw = int(nworkers)
        self.conns, self.procs = [], []
        for _ in range(self.nw):
            a, b = ctx.Pipe()
            p = ctx.Process(target=_worker, args=(b,), daemon=True)
            p.start()
            self.conns.append(a), self.procs.append(p)
        self.shm = []
        self.X = self.Y = None

    def _all(self, msgs):
        for c, m in zip(self.conns, msgs):
            c.send(m)
        out = []
        for c in self.conns:
            r = c.recv()
            if r[0] != "ok":
                raise RuntimeError("direct_pool worker: %s" % (r[1],))
            out.append(r[1])
        return out

    def factor(self, Kcsc):
        Kcsc = Kcsc.tocsc()
        Kcsc.sort_indices()
        msg = ("factor", Kcsc.shape[0], np.asarray(Kcsc.indptr), np.asarray(Kcsc.indices), np.asarray(Kcsc.data))
        return self._all([msg] * self.nw)[0]

    def attach(self, n, ncol):
        nbytes = 8 * n * ncol
        self.shm = [shared_memory.SharedMemory(create=True, size=nbytes), shared_memory.SharedMemory(create=True, size=nbytes)]
        self.X = np.ndarray((ncol, n), dtype=np.float64, buffer=self.shm[0].buf)
        self.Y = np.ndarray((ncol, n), dtype=np.float64, buffer=self.shm[1].buf)
        self.ncol = ncol
        self._all([("attach", self.shm[0].name, self.shm[1].name, n, ncol)] * self.nw)
        cuts = [(ncol * w) // self.nw for w in range(self.nw + 1)]
        self.ranges = [(cuts[w], cuts[w + 1]) for w in range(self.nw)]

    def solve(self):
        """Y[c] = K^{-1} X[c] for every column c (X, Y: the shared (ncol, n) arrays)."""
        self._all([("solve", a, b) for a, b in self.ranges])

    def close(self):
        for c in self.conns:
            try:
                c.send(("stop",))
            except (OSError, BrokenPipeError):
                pass
        for p in self.procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        for s in self.shm:
            try:
                s.close()
                s.unlink()
            except FileNotFoundError:
                pass
        self.shm = []
