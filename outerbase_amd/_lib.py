"""ctypes binding of libobhip.so (the C ABI declared in include/obhip.h).

The prototypes are read from the header itself, so the Python side can never
drift from the ABI.  There is no fallback of any kind: if the shared library is
missing, importing this module raises, and every device-touching call fails
with OBHIP_ERR_NO_DEVICE when no gfx950 GPU is visible.
"""
import ctypes as C
import os
import re

# torch bundles its own HIP runtime; it must be the first (and only) libamdhip64
# in the process, otherwise device enumeration fails in whichever library
# initialised second.  torch is this package's device-memory / collective
# plumbing anyway.
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover - host-only use (model, term selection)
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "obhip.h")
LIB_PATH = os.path.join(_HERE, "libobhip.so")
# tests/fault_inject_worker.py (a child process of one test) loads the test build instead: the
# same sources with the exchange's fault injector compiled in (csrc/Makefile, OBHIP_TESTING)
if os.environ.get("OBHIP_TEST_LIBRARY") == "testing":
    LIB_PATH = os.path.join(_HERE, "libobhip_testing.so")

# obhip_host_allreduce_fn: sum count doubles of a HOST buffer in place over all ranks
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64)

_BASE = {
    "int": C.c_int, "double": C.c_double, "uint64_t": C.c_uint64,
    "int64_t": C.c_int64, "uint32_t": C.c_uint32, "uint16_t": C.c_uint16, "void": None, "char": C.c_char, "size_t": C.c_size_t,
}
_HANDLES = ("obhip_model", "obhip_basis", "obhip_terms", "obhip_comm", "obhip_lpdf",
            "obhip_predictor")


def _ctype(decl):
    decl = decl.replace("const", " ").strip()
    stars = decl.count("*")
    base = decl.replace("*", " ").split()[0]
    if base == "obhip_host_allreduce_fn":
        return C.c_void_p  # pass ctypes.cast(HOST_ALLREDUCE_FN(f), c_void_p) or None
    if base in _HANDLES:
        return C.c_void_p if stars == 1 else C.POINTER(C.c_void_p)
    if base == "char" and stars == 1:
        return C.c_char_p
    if base == "void":
        if stars == 0:
            return None
        return C.c_void_p if stars == 1 else C.POINTER(C.c_void_p)
    t = _BASE[base]
    if stars == 0:
        return t
    # data pointers are passed as raw addresses (numpy / torch / device)
    return C.c_void_p


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function in obhip.h."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    src = re.sub(r'extern\s+"C"\s*\{', " ", src)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(obhip_\w+)\s*\(([^;{}]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        argt = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name
                a = re.sub(r"\b\w+\s*$", "", a) if re.search(r"[\*\s]\w+\s*$", a) else a
                argt.append(_ctype(a))
        protos[name] = (_ctype(ret), argt)
    return protos


class ObhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("obhip error %d: %s" % (code, msg))
        self.code = code


if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libobhip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; "
        "g.build()'` or `make -C outerbase_amd/csrc`. There is no CPU fallback." % LIB_PATH)

lib = C.CDLL(LIB_PATH)
PROTOS = parse_header()
for _name, (_res, _args) in PROTOS.items():
    _f = getattr(lib, _name)
    _f.restype = _res
    _f.argtypes = _args


def check(rc):
    if rc != 0:
        raise ObhipError(rc, lib.obhip_last_error().decode("utf-8", "replace"))


def call(name, *args):
    check(getattr(lib, name)(*args))


def ptr(a):
    """Address of a numpy array / torch tensor / int / None."""
    if a is None:
        return None
    if isinstance(a, int):
        return a
    if hasattr(a, "data_ptr"):
        return a.data_ptr()
    return a.ctypes.data


def device_count():
    n = C.c_int(0)
    call("obhip_device_count", C.byref(n))
    return n.value
