"""ctypes wrapper of oracle/_build/libob_cpu.so (oracle/ob_cpu.cpp) -- TEST
INFRASTRUCTURE ONLY (second CPU implementation for cross-checks and the
cpu_baseline leg of bench.py).  Build with `make -C oracle`."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libob_cpu.so")
KIND_ID = {"mat25": 0, "mat25pow": 1, "mat25ang": 2}


def available():
    return os.path.exists(_PATH)


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_PATH)
        _lib.ob_cpu_num_procs.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def build(om, x, nthreads=0):
    """outerbase::build with the reference's chunk schedule -> (basemat, basescale)."""
    x = np.asfortranarray(x, dtype=np.float64)
    n, d = x.shape
    M = len(om.knotpt)
    kinds = np.array([KIND_ID[k] for k in om.kinds], dtype=np.int32)
    st = np.ascontiguousarray(om.knotptst, dtype=np.uint64)
    hst = np.ascontiguousarray(om.hypst, dtype=np.uint64)
    rot = np.asfortranarray(om.rotmat)
    basemat = np.empty((n, M), order="F")
    basescale = np.empty(n)
    lib().ob_cpu_build(C.c_int64(n), C.c_int64(d), _p(x), _p(kinds), _p(st),
                       _p(np.ascontiguousarray(om.knotpt)), _p(hst),
                       _p(np.ascontiguousarray(om.hyp)), _p(rot), C.c_int64(rot.shape[0]),
                       C.c_int(nthreads), _p(basemat), _p(basescale))
    return basemat, basescale


def _terms(terms):
    return np.asfortranarray(np.asarray(terms).astype(np.uint64))


def getmat(om, terms, basemat, basescale, nthreads=0):
    t = _terms(terms)
    n = basemat.shape[0]
    out = np.empty((n, t.shape[0]), order="F")
    lib().ob_cpu_getmat(C.c_int64(n), C.c_int64(t.shape[0]), C.c_int64(t.shape[1]), _p(t),
                        _p(np.ascontiguousarray(om.knotptst, dtype=np.uint64)), _p(basemat),
                        _p(basescale), C.c_int(nthreads), _p(out))
    return out


def mm(om, terms, basemat, basescale, a, nthreads=0):
    t = _terms(terms)
    n = basemat.shape[0]
    out = np.empty(n)
    a = np.ascontiguousarray(a, dtype=np.float64)
    lib().ob_cpu_mm(C.c_int64(n), C.c_int64(t.shape[0]), C.c_int64(t.shape[1]), _p(t),
                    _p(np.ascontiguousarray(om.knotptst, dtype=np.uint64)), _p(basemat),
                    _p(basescale), _p(a), C.c_int(nthreads), _p(out))
    return out


def tmm(om, terms, basemat, basescale, a, nthreads=0):
    t = _terms(terms)
    n = basemat.shape[0]
    out = np.empty(t.shape[0])
    a = np.ascontiguousarray(a, dtype=np.float64)
    lib().ob_cpu_tmm(C.c_int64(n), C.c_int64(t.shape[0]), C.c_int64(t.shape[1]), _p(t),
                     _p(np.ascontiguousarray(om.knotptst, dtype=np.uint64)), _p(basemat),
                     _p(basescale), _p(a), C.c_int(nthreads), _p(out))
    return out
