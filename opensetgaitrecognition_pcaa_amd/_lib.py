"""ctypes binding of the C ABI declared in ``include/pcaa_hip.h``.

Prototypes are parsed from the header itself, so the Python binding cannot
drift from the declared ABI.  There is NO fallback: if ``libpcaa_hip.so`` is
missing the import of any compute entry point raises with build instructions.
"""
import ctypes
import os
import re
import threading

PKG = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(PKG), "include", "pcaa_hip.h")
LIB_PATH = os.path.join(PKG, "libpcaa_hip.so")

PCAA_F32, PCAA_BF16 = 0, 1


def _header_abi_version():
    import re
    hdr = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "pcaa_hip.h")
    with open(hdr) as f:
        return int(re.search(r"#define\s+PCAA_ABI_VERSION\s+(\d+)", f.read()).group(1))


ABI_VERSION = _header_abi_version()        # what include/pcaa_hip.h declares; load() checks the library against it
KC, RC = 0, 1
ACT_NONE, ACT_ELU = 0, 1

_lock = threading.Lock()
_lib = None


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every function the header declares."""
    with open(path) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#.*$", " ", src, flags=re.M)
    src = src.replace('extern "C" {', " ").replace("}", " ")
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(pcaa_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        protos[name] = (_ctype(ret), [] if args in ("", "void") else [_ctype(_strip_name(a)) for a in args.split(",")])
    return protos


def _strip_name(arg):
    arg = " ".join(arg.split())
    m = re.match(r"^(.*?)(\b\w+)$", arg)
    base = m.group(1).strip() if m and m.group(1).strip() else arg
    return base


def _ctype(t):
    t = " ".join(t.replace("*", " * ").split())
    if "*" in t:
        return ctypes.c_char_p if t == "const char *" else ctypes.c_void_p
    return {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
            "size_t": ctypes.c_size_t, "void": None, "double": ctypes.c_double}[t]


def load():
    """Load (once) and return the ctypes library with argtypes set."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the PCAA HIP extension has not been built. "
                "Run `python -m opensetgaitrecognition_pcaa_amd.build` (needs hipcc). "
                "There is no CPU fallback for this package.")
        # torch first: it ships its own libamdhip64, and the HIP runtime that owns the device context and the
        # streams we launch on must be the one libpcaa_hip.so binds to.  Loaded the other way round (this
        # library before torch, e.g. build() then smoke() in one process) the process ends up with two HIP
        # runtimes and every launch fails with "no ROCm-capable device is detected".
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (ret, args) in parse_header().items():
            fn = getattr(lib, name)     # AttributeError if the .so does not export it
            fn.restype = ret
            fn.argtypes = args
        # libpcaa_hip.so is git-ignored and travels prebuilt: a stale one would export the same names with other
        # signatures and be called with mismatched arguments
        built = lib.pcaa_abi_version()
        if built != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} was built for ABI version {built}, include/pcaa_hip.h declares "
                               f"{ABI_VERSION}: rebuild it (python -m opensetgaitrecognition_pcaa_amd.build)")
        _lib = lib
    return _lib


class PcaaError(RuntimeError):
    pass


def check(rc, what=""):
    if rc != 0:
        msg = load().pcaa_last_error()
        raise PcaaError(f"{what}: error {rc}: {msg.decode() if msg else ''}")
