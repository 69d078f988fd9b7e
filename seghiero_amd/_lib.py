"""ctypes binding of libseghiero_hip.so (the C ABI declared in include/seghiero_hip.h).

The prototypes are parsed from the header itself, so the Python side can never drift from the
declared ABI, and `exported_symbols()` lets the CPU test-suite check that the library exports every
declared entry point.  There is NO fallback: if the library is missing or a call returns an error
the product raises.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "seghiero_hip.h")
LIBPATH = os.environ.get("SEGHIERO_LIB") or os.path.join(_HERE, "libseghiero_hip.so")      # SEGHIERO_LIB: kernel experiments

_SCALARS = {
    "int": ctypes.c_int, "int64_t": ctypes.c_longlong, "long long": ctypes.c_longlong,
    "float": ctypes.c_float, "double": ctypes.c_double,
}


class SegHieroHipError(RuntimeError):
    pass


# status codes of include/seghiero_hip.h
STATUS = {0: "ok", -1: "SH_EINVAL: invalid argument", -2: "SH_ELAUNCH: HIP launch error",
          -3: "SH_EUNSUPPORTED: no kernel instantiation for this geometry, nothing was launched"}


def status_text(rc):
    return STATUS.get(rc, "unknown status")


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    text = text.replace('extern "C" {', " ").replace("}", " ")
    protos = {}
    for m in re.finditer(r"\b(int64_t|int)\s+(sh_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        argl = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argl.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, nm = a.rsplit(" ", 1)
                    ty = ty.replace("const", "").strip()
                    argl.append((_SCALARS[ty], nm))
        protos[name] = (_SCALARS[ret], argl)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self._fn = {}
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIBPATH):
                raise SegHieroHipError(
                    f"{LIBPATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(seghiero_amd has no CPU or PyTorch fallback)")
            self._dll = ctypes.CDLL(LIBPATH)
            for name, (ret, args) in self.protos.items():
                f = getattr(self._dll, name)     # AttributeError if the .so lacks a declared symbol
                f.restype = ret
                f.argtypes = [t for t, _ in args]
                self._fn[name] = f
        return self

    def exported_symbols(self):
        self.load()
        return sorted(self._fn)

    def raw(self, name):
        self.load()
        return self._fn[name]

    def call(self, name, *args):
        """Call an int-returning entry point; raise on a non-zero status."""
        rc = self.raw(name)(*args)
        if rc != 0:
            raise SegHieroHipError(f"{name} failed with status {rc} ({status_text(rc)})")
        return rc


LIB = _Lib()
