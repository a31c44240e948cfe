"""ctypes binding of the C ABI declared in include/llmrec_hip.h.

The prototypes are parsed from the header itself, so the Python argtypes cannot drift from the
C declarations. There is deliberately NO fallback: if libllmrec_hip.so is missing or a call
returns a non-zero status, a RuntimeError is raised (the product path must fail loudly rather
than silently run something else)."""
from __future__ import annotations

import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "llmrec_hip.h")
# LLMREC_LIB: tools/ load the instrumented build (python -m llmrec_amd.build --tools) through the same binding
LIB_PATH = os.environ.get("LLMREC_LIB") or os.path.join(HERE, "lib", "libllmrec_hip.so")

_SCALARS = {
    "int": ctypes.c_int, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64, "uint32_t": ctypes.c_uint32, "float": ctypes.c_float,
    "double": ctypes.c_double, "llmrec_stream_t": ctypes.c_void_p, "void": None,
}


def parse_header(path: str = HEADER):
    """Return {name: (restype, [argtypes], [argnames])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"#[^\n]*", " ", text)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(llmrec_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        restype = ctypes.c_char_p if "char" in ret and "*" in ret else _SCALARS[ret.replace("const", "").strip()]
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                argnames.append(re.findall(r"\w+", a)[-1])
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_SCALARS[a.replace("const", "").split()[0]])
        protos[name] = (restype, argtypes, argnames)
    return protos


def header_constants(path: str = HEADER):
    out = {}
    for m in re.finditer(r"#define\s+(LLMREC_\w+)\s+(-?\d+)", open(path).read()):
        out[m.group(1)] = int(m.group(2))
    return out


CONST = header_constants()
_lib = None
_protos = None


def load():
    """Load the shared library (once) and attach the parsed prototypes."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    # torch bundles its own ROCm runtime (torch/lib/libamdhip64.so); it must be the one in the
    # process before this library resolves libamdhip64.so.7, or two HIP runtimes get loaded and
    # launches fail with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "llmrec_amd: %s is missing. Build it with `python -m llmrec_amd.build` (hipcc, gfx950). "
            "There is no CPU fallback for the HIP hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (restype, argtypes, _) in _protos.items():
        fn = getattr(lib, name)            # AttributeError here = header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.llmrec_abi_version() != CONST["LLMREC_ABI_VERSION"]:
        raise RuntimeError("llmrec_amd: libllmrec_hip.so ABI %d != header ABI %d; rebuild"
                           % (lib.llmrec_abi_version(), CONST["LLMREC_ABI_VERSION"]))
    _lib = lib
    return lib


_TRACE = os.environ.get("LLMREC_TRACE_CALLS", "0") == "1"
n_calls = 0      # entry-point invocations so far (FusedStep reports the per-step delta: a launch-count proxy without a profiler)


def call(name: str, *args):
    """Invoke an int-returning entry point; raise on a non-zero status."""
    global n_calls
    n_calls += 1
    lib = load()
    if _TRACE:                                               # LLMREC_TRACE_CALLS=1 (debugging a device fault): name each entry point, synchronise behind it
        import sys
        import torch
        print("[llmrec] %s" % name, file=sys.stderr, flush=True)
        status = getattr(lib, name)(*args)
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.synchronize()
    else:
        status = getattr(lib, name)(*args)
    if status != 0:
        raise RuntimeError("%s failed: %s (%s)" % (
            name, lib.llmrec_status_string(status).decode(), lib.llmrec_last_error().decode()))


def query(name: str, *args) -> int:
    """Invoke an int64-returning size query."""
    return int(getattr(load(), name)(*args))
