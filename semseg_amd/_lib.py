"""ctypes binding of libsemseg_hip.so.  Prototypes are parsed from include/semseg_hip.h so the header
is the single source of truth for the C ABI.  There is NO fallback: if the library is missing or a
symbol is absent, importing/using the ops raises."""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "semseg_hip.h")
def debug(name, default=None):
    """Value (a string) of the A/B switch `name` in the environment variable SEMSEG_DEBUG = "name=value,name=value,...", or
    `default`.  Everything a measurement or a test toggles and a user never needs lives there (DESIGN.md section 10 lists the
    names); what a user may set has a variable of its own: SEMSEG_ARITH, SEMSEG_SYNCBN_XCHG, SEMSEG_STEP_PLAN,
    SEMSEG_CHECK_LABELS, SEMSEG_HIP_LIB (SEMSEG_FORCE_DIST / SEMSEG_TILE_TUNE: test and tuning drivers)."""
    for item in os.environ.get("SEMSEG_DEBUG", "").split(","):
        k, _, v = item.partition("=")
        if k.strip() == name:
            return v.strip()
    return default


LIB_PATH = os.environ.get("SEMSEG_HIP_LIB") or os.path.join(HERE, "csrc", "libsemseg_hip.so")  # override: kernel tuning only

_CT = {
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double,
    "size_t": ctypes.c_size_t, "hipStream_t": ctypes.c_void_p, "long long": ctypes.c_longlong, "unsigned long long": ctypes.c_ulonglong,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])}"""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|size_t)\s+(semseg_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        alist = []
        for a in args.split(","):
            a = " ".join(a.split())
            if not a or a == "void":
                continue
            if "*" in a:
                ct = ctypes.c_void_p
                an = a.split("*")[-1].strip()
            else:
                parts = a.split()
                an = parts[-1]
                ty = " ".join(p for p in parts[:-1] if p != "const")
                ct = _CT[ty]
            alist.append((ct, an))
        protos[name] = (_CT[ret], alist)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self._fn = {}
        self._argtypes = {}
        self.recorder = None      # semseg_amd.plan.StepPlan while it records: every successful call is appended to it

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "libsemseg_hip.so not built (%s). Run `python -c 'import __graft_entry__ as g; "
                    "g.build()'` or `python -m semseg_amd.build`. There is no CPU fallback." % LIB_PATH)
            # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's): it must be the one already
            # resident when our library resolves its HIP dependency, otherwise two HIP runtimes end up in
            # the process and every launch on a torch stream fails
            import torch  # noqa: F401
            self._dll = ctypes.CDLL(LIB_PATH)
            for name, (ret, args) in parse_header().items():
                f = getattr(self._dll, name)  # AttributeError if the symbol is missing
                f.restype = ret
                f.argtypes = [a[0] for a in args]
                self._fn[name] = f
                self._argtypes[name] = f.argtypes
        return self

    def __getattr__(self, name):
        self.load()
        try:
            f = self._fn[name]
        except KeyError:
            raise AttributeError(name)
        r = self.recorder
        return f if r is None else r.wrap(name, f, self._argtypes[name])

    def raw(self, name):
        """The ctypes function itself, never the recording wrapper."""
        self.load()
        return self._fn[name]


lib = _Lib()
