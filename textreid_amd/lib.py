"""ctypes binding of libtextreid_hip.so (the C ABI declared in include/textreid_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  Loading
fails loudly when the shared object is missing, and every entry point raises
``RuntimeError`` with the library's message when a call is rejected.
"""

import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TRID_LIB_PATH") or os.path.join(_HERE, "libtextreid_hip.so")  # override: kernel experiments

TRID_A_KC, TRID_A_MC, TRID_A_CONV = 0, 1, 2
TRID_B_KC, TRID_B_NC, TRID_B_CONV = 0, 1, 2


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", ctypes.c_void_p),
        ("B", ctypes.c_void_p),
        ("C", ctypes.c_void_p),
        ("M", ctypes.c_int32),
        ("N", ctypes.c_int32),
        ("K", ctypes.c_int32),
        ("lda", ctypes.c_int64),
        ("ldb", ctypes.c_int64),
        ("ldc", ctypes.c_int64),
        ("strideA", ctypes.c_int64),
        ("strideB", ctypes.c_int64),
        ("strideC", ctypes.c_int64),
        ("batch", ctypes.c_int32),
        ("splits", ctypes.c_int32),
        ("strideSplit", ctypes.c_int64),
        ("a_mode", ctypes.c_int32),
        ("b_mode", ctypes.c_int32),
        ("alpha", ctypes.c_float),
        ("accumulate", ctypes.c_int32),
        ("bias", ctypes.c_void_p),
        ("strideBias", ctypes.c_int64),
        ("stats", ctypes.c_void_p),
        ("H", ctypes.c_int32),
        ("W", ctypes.c_int32),
        ("Cin", ctypes.c_int32),
        ("precision", ctypes.c_int32),
        ("residual", ctypes.c_void_p),
        ("ldres", ctypes.c_int64),
        ("relu", ctypes.c_int32),
        ("a_amax", ctypes.c_void_p),
        ("b_amax", ctypes.c_void_p),
        ("stats_minmax", ctypes.c_int32),
        ("c_format", ctypes.c_int32),
        ("c_mask", ctypes.c_void_p),
        ("col_scale", ctypes.c_void_p),
        ("res_p16", ctypes.c_void_p),
        ("res_amax", ctypes.c_void_p),
        ("eval_coef", ctypes.c_void_p),
        ("eval_tin", ctypes.c_void_p),
        ("eval_tres", ctypes.c_void_p),
        ("out_bound", ctypes.c_void_p),
        ("out_tmax", ctypes.c_void_p),
        ("eval_pool_w", ctypes.c_int32),
        ("bnb_y", ctypes.c_void_p),
        ("bnb_mean", ctypes.c_void_p),
        ("bnb_invstd", ctypes.c_void_p),
        ("bnb_scale", ctypes.c_void_p),
        ("bnb_shift", ctypes.c_void_p),
        ("bnb_ws", ctypes.c_void_p),
        ("bnb_ws2", ctypes.c_void_p),
        ("bnb_relu", ctypes.c_int32),
        ("bnb_mask", ctypes.c_void_p),
    ]


_T = {"p": ctypes.c_void_p, "i": ctypes.c_int, "l": ctypes.c_longlong, "f": ctypes.c_float}
_RET = {"int": ctypes.c_int, "long long": ctypes.c_longlong, "const char*": ctypes.c_char_p}

HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "textreid_hip.h")


def _arg_code(arg):
    arg = arg.strip()
    if arg in ("void", ""):
        return ""
    if "*" in arg:
        return "p"
    if "long long" in arg or "int64_t" in arg:
        return "l"
    if "float" in arg:
        return "f"
    if "int" in arg:
        return "i"
    raise ValueError("unhandled C argument type: %r" % arg)


def parse_header(path=HEADER_PATH):
    """{name: (return type, arg codes)} for every function include/textreid_hip.h
    declares -- the single source of truth for the binding."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"typedef struct.*?\}\s*\w+;", "", text, flags=re.S)
    text = re.sub(r"enum\s*\{.*?\};", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"(const char\*|long long|int)\s+(trid_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        out[name] = (ret, "".join(_arg_code(a) for a in args.split(",")))
    return out


DECLS = parse_header()
EXPORTS = sorted(DECLS)

_lib = None


class HipExtensionMissing(RuntimeError):
    pass


def load():
    """Load the shared library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            "libtextreid_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C textreid_amd/csrc`. There is no CPU fallback." % LIB_PATH
        )
    import torch  # noqa: F401  (load PyTorch's HIP runtime first so both share one libamdhip64)

    lib = ctypes.CDLL(LIB_PATH)
    for name, (ret, sig) in DECLS.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = _RET[ret]
        fn.argtypes = [_T[c] for c in sig]
    _lib = lib
    return lib


def last_error():
    return load().trid_last_error_string().decode("utf-8", "replace")


# Optional call log (tools/step_calls.py): a list -> every entry-point call appends (name, scalar arguments, start event, end
# event) on torch's current stream.  None in the product path.
TRACE = None


def _traced(name, args):
    import torch

    scal = tuple(a for a in args if isinstance(a, (int, float)) and not isinstance(a, bool) and abs(a) < (1 << 32))
    if name in ("trid_gemm_f32", "trid_gemm_p16", "trid_gemm_p16_wgrad"):
        d = GemmDesc.from_address(args[0])
        scal = tuple("%s=%s" % (f, getattr(d, f)) for f in ("a_mode", "b_mode", "M", "N", "K", "batch", "splits", "accumulate", "precision", "H", "W", "Cin") if hasattr(d, f))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = getattr(load(), name)(*args)
    e1.record()
    TRACE.append((name, scal, e0, e1))
    return rc


def call(name, *args):
    """Invoke a status-returning entry point; raise RuntimeError on failure."""
    rc = getattr(load(), name)(*args) if TRACE is None else _traced(name, args)
    if rc != 0:
        raise RuntimeError("%s failed (rc=%d): %s" % (name, rc, last_error()))
