"""Thin object layer over the C ABI: context, device vectors, CSR matrices, operators.

Device memory is owned by libpermonhip (hipMalloc); numpy arrays cross the boundary only in
Vec.from_numpy / Vec.to_numpy.  Nothing here computes: every method is one C-ABI call.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import check

EPS = float(np.finfo(np.float64).eps)


class Context:
    """One GPU (one rank).  Mirrors what PermonInitialize gives a rank: a device and a communicator."""

    def __init__(self, device=0):
        self.L = _lib.load(strict=not os.environ.get("PMH_PARTIAL_ABI"))
        h = C.c_void_p()
        check(self.L.pmh_init(int(device), C.byref(h)))
        self.h = h
        self.device = int(device)
        self.rank, self.size = 0, 1

    def close(self):
        if self.h:
            self.L.pmh_finalize(self.h)
            self.h = None

    def name(self):
        buf = C.create_string_buffer(256)
        check(self.L.pmh_device_name(self.h, buf, 256))
        return buf.value.decode()

    def sync(self):
        check(self.L.pmh_sync(self.h))

    def mem_info(self):
        """(free, total) bytes of the device's HBM."""
        f, t = C.c_size_t(), C.c_size_t()
        check(self.L.pmh_mem_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def timer_start(self):
        check(self.L.pmh_timer_start(self.h))

    def timer_stop(self):
        ms = C.c_double()
        check(self.L.pmh_timer_stop(self.h, C.byref(ms)))
        return ms.value

    # ---- RCCL communicator (one process per GPU) --------------------------------------------------
    def comm_unique_id(self):
        buf = (C.c_ubyte * 128)()
        check(self.L.pmh_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, rank, size, unique_id):
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        check(self.L.pmh_comm_init(self.h, int(rank), int(size), buf))
        self.rank, self.size = int(rank), int(size)

    def comm_set_host_transport(self, rank, size, allreduce):
        """pmh_comm_set_host_transport: every collective of the data path goes through `allreduce(op, array)` -- an IN-PLACE all-reduce over the ranks of a float64 numpy
        view of the pinned staging buffer (op 0 sum, 1 min, 2 barrier with an empty array) -- instead of RCCL.  allreduce = None removes it."""
        if allreduce is None:
            check(self.L.pmh_comm_set_host_transport(self.h, 0, 1, None, None))
            self._host_cb = None
            self.rank, self.size = 0, 1
            return
        import numpy as np

        proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_size_t)

        def cb(_user, op, buf, count):
            try:
                arr = np.ctypeslib.as_array(buf, shape=(count,)) if count else np.empty(0)
                allreduce(int(op), arr)
                return 0
            except Exception:  # noqa: BLE001 - reported through the library's error channel
                import traceback

                traceback.print_exc()
                return 1

        self._host_cb = proto(cb)  # kept alive with the context
        check(self.L.pmh_comm_set_host_transport(self.h, int(rank), int(size), C.cast(self._host_cb, C.c_void_p), None))
        self.rank, self.size = int(rank), int(size)

    def comm_rank(self):
        """(rank, size) of the RCCL communicator (pmh_comm_rank); (0, 1) without one."""
        r, n = C.c_int(), C.c_int()
        check(self.L.pmh_comm_rank(self.h, C.byref(r), C.byref(n)))
        return r.value, n.value

    def barrier(self):
        check(self.L.pmh_comm_barrier(self.h))

    # ---- vectors -----------------------------------------------------------------------------------
    def vec(self, n):
        return Vec(self, n)

    def vec_from(self, a):
        return Vec.from_numpy(self, a)


class Vec:
    """fp64 device vector (PETSc Vec role).  `.p` is the raw device pointer handed to the C ABI."""

    def __init__(self, ctx, n, zero=True):
        self.ctx, self.n = ctx, int(n)
        p = C.c_void_p()
        check(ctx.L.pmh_malloc(ctx.h, 8 * max(self.n, 1), C.byref(p)))
        self.p = p
        if zero:
            check(ctx.L.pmh_memset(ctx.h, self.p, 0, 8 * self.n))

    @classmethod
    def borrowed(cls, ctx, p, n):
        """View of device memory owned by a library object (never freed from here)."""
        v = cls.__new__(cls)
        v.ctx, v.n, v.p, v._borrowed = ctx, int(n), C.c_void_p(p) if not isinstance(p, C.c_void_p) else p, True
        return v

    @classmethod
    def from_numpy(cls, ctx, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        v = cls(ctx, a.size, zero=False)
        check(ctx.L.pmh_memcpy_h2d(ctx.h, v.p, a.ctypes.data_as(C.c_void_p), 8 * a.size))
        return v

    def set_numpy(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.n
        check(self.ctx.L.pmh_memcpy_h2d(self.ctx.h, self.p, a.ctypes.data_as(C.c_void_p), 8 * a.size))

    def to_numpy(self):
        a = np.empty(self.n, dtype=np.float64)
        check(self.ctx.L.pmh_memcpy_d2h(self.ctx.h, a.ctypes.data_as(C.c_void_p), self.p, 8 * self.n))
        return a

    def copy(self):
        v = Vec(self.ctx, self.n, zero=False)
        check(self.ctx.L.pmh_memcpy_d2d(self.ctx.h, v.p, self.p, 8 * self.n))
        return v

    def free(self):
        if self.p and not getattr(self, "_borrowed", False):
            self.ctx.L.pmh_free(self.ctx.h, self.p)
        self.p = None

    # PETSc Vec ops used by the path
    def axpy(self, a, x):
        check(self.ctx.L.pmh_vec_axpy(self.ctx.h, self.n, self.p, float(a), x.p))

    def aypx(self, a, x):
        check(self.ctx.L.pmh_vec_aypx(self.ctx.h, self.n, self.p, float(a), x.p))

    def waxpy(self, a, x, y):
        check(self.ctx.L.pmh_vec_waxpy(self.ctx.h, self.n, self.p, float(a), x.p, y.p))

    def scale(self, a):
        check(self.ctx.L.pmh_vec_scale(self.ctx.h, self.n, self.p, float(a)))

    def set(self, a):
        check(self.ctx.L.pmh_vec_set(self.ctx.h, self.n, self.p, float(a)))

    def dot(self, y):
        r = C.c_double()
        check(self.ctx.L.pmh_vec_dot(self.ctx.h, self.n, self.p, y.p, C.byref(r)))
        return r.value

    def norm(self):
        r = C.c_double()
        check(self.ctx.L.pmh_vec_norm2(self.ctx.h, self.n, self.p, C.byref(r)))
        return r.value


def _ptr(v):
    return v.p if v is not None else None


class CsrMat:
    """PETSc SeqAIJ role: CSR with int32 indices and fp64 values resident in HBM."""

    def __init__(self, ctx, nrows, ncols, rowptr, col, val):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        if rowptr.size != nrows + 1:
            raise ValueError("rowptr must have nrows+1 entries")
        if col.size != val.size or (nrows and col.size != rowptr[-1]):
            raise ValueError("col/val length does not match rowptr[-1]")
        self.ctx, self.nrows, self.ncols, self.nnz = ctx, int(nrows), int(ncols), int(col.size)
        h = C.c_void_p()
        check(ctx.L.pmh_csr_create(ctx.h, self.nrows, self.ncols, rowptr.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p),
                                   val.ctypes.data_as(C.c_void_p), C.byref(h)))
        self.h = h

    def mult(self, x, y):  # MatMult
        check(self.ctx.L.pmh_csr_mult(self.h, x.p, y.p))

    def mult_add(self, x, y1, y):  # MatMultAdd
        check(self.ctx.L.pmh_csr_mult_add(self.h, x.p, y1.p, y.p))

    def mult_transpose(self, x, y):  # MatMultTranspose
        check(self.ctx.L.pmh_csr_mult_transpose(self.h, x.p, y.p))

    def algorithmic_bytes(self):
        b = C.c_double()
        check(self.ctx.L.pmh_csr_algorithmic_bytes(self.h, C.byref(b)))
        return b.value

    def timing_enable(self, max_launches):
        check(self.ctx.L.pmh_csr_timing_enable(self.h, int(max_launches)))

    def timing_get(self, epilogue):
        """(launches, total milliseconds) of the SpMV launches with the given epilogue since timing_enable."""
        n, ms = C.c_int(), C.c_double()
        check(self.ctx.L.pmh_csr_timing_get(self.h, int(epilogue), C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def destroy(self):
        if self.h:
            self.ctx.L.pmh_csr_destroy(self.h)
            self.h = None


class Op:
    """A Mat with a mult slot (PETSc Mat role in the QP chain)."""

    def __init__(self, ctx, h, n, keep=()):
        self.ctx, self.h, self.n, self._keep = ctx, h, int(n), list(keep)

    @classmethod
    def from_csr(cls, A):
        h = C.c_void_p()
        check(A.ctx.L.pmh_op_create_csr(A.h, C.byref(h)))
        return cls(A.ctx, h, A.nrows, keep=[A])

    @classmethod
    def shell(cls, ctx, n, fn):
        """fn(x_dev_ptr, y_dev_ptr) -> None, both raw device pointers (MatCreateShellPermon role)."""

        def _cb(user, xp, yp):
            try:
                fn(C.c_void_p(xp), C.c_void_p(yp))
                return 0
            except Exception:  # noqa: BLE001 - reported through the C error path
                import traceback

                traceback.print_exc()
                return 1

        cb = _lib.SHELL_MULT_FN(_cb)
        h = C.c_void_p()
        check(ctx.L.pmh_op_create_shell(ctx.h, int(n), cb, None, C.byref(h)))
        return cls(ctx, h, n, keep=[cb, fn])

    def mult(self, x, y):
        check(self.ctx.L.pmh_op_mult(self.h, x.p, y.p))

    def max_eigenvalue(self, tol=-1.0, maxits=-1):
        lam, its = C.c_double(), C.c_int()
        check(self.ctx.L.pmh_op_max_eigenvalue(self.h, float(tol), int(maxits), C.byref(lam), C.byref(its)))
        return lam.value, its.value

    def destroy(self):
        if self.h:
            self.ctx.L.pmh_op_destroy(self.h)
            self.h = None
