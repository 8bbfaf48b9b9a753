"""ctypes binding of libvdn_render.so, generated from include/vdn_render.h at import time.

The header is the single source of truth: struct layouts and entry points are parsed from it, so
the Python mirror cannot drift from the C ABI. There is no fallback: if the shared library is
missing or does not load, every kernel entry point raises.
"""
import ctypes
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
HEADER = os.path.join(os.path.dirname(PKG), "include", "vdn_render.h")
# (VDN_LIB: another build of the same library, for A/B timing of compile-time switches inside one gpurun call; development only)
LIB_PATH = os.environ.get("VDN_LIB") or os.path.join(HERE, "libvdn_render.so")

_SCALARS = {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "int": ctypes.c_int, "double": ctypes.c_double}


def _strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def parse_header(path=HEADER):
    """-> (structs: {name: [(field, ctype)]}, functions: {name: [argument ctypes]})"""
    text = _strip_comments(open(path).read())
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*(.+)$", decl)
            base, ptr, names = mm.group(2), mm.group(3), mm.group(4)
            for nm in names.split(","):
                nm = nm.strip()
                is_ptr = bool(ptr) or nm.startswith("*")
                nm = nm.lstrip("* ")
                fields.append((nm, ctypes.c_void_p if is_ptr else _SCALARS[base]))
        structs[m.group(2)] = fields
    funcs = {}
    for m in re.finditer(r"\bint\s+(vdn_\w+)\s*\((.*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        types = []
        for a in ([] if args in ("", "void") else args.split(",")):
            if "*" in a:
                types.append(ctypes.c_void_p)          # struct pointers, device pointers and the hipStream_t handle
            else:
                types.append(next(t for n, t in _SCALARS.items() if re.search(r"\b%s\b" % n, a)))
        funcs[m.group(1)] = types
    return structs, funcs


_STRUCT_FIELDS, FUNCTIONS = parse_header()


def _make_struct(name, fields):
    return type(name, (ctypes.Structure,), {"_fields_": fields})


STRUCTS = {n: _make_struct(n, f) for n, f in _STRUCT_FIELDS.items()}
globals().update(STRUCTS)


def struct_dtype(name):
    """numpy dtype with the C layout of struct `name` (for device-side descriptor tables)."""
    return np.dtype(STRUCTS[name])


class VdnError(RuntimeError):
    pass


_lib = None


def load():
    """Load the kernel library, failing loudly when it is absent (no CPU / eager fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: the library must bind to the HIP runtime torch has loaded (its own copy of libamdhip64); loaded the other
    # way round the process ends up with two runtimes and every launch fails with hipErrorNoDevice
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise VdnError("HIP kernel library %s not built. Run `python -m vdn_hip.build` (or "
                       "__graft_entry__.build()); there is no fallback path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for fn, argtypes in FUNCTIONS.items():
        f = getattr(lib, fn)       # AttributeError here = header/library mismatch
        f.restype = ctypes.c_int
        f.argtypes = argtypes      # from the header: a 64-bit stream handle must not be narrowed to a C int
    if lib.vdn_abi_version() != int(re.search(r"#define\s+VDN_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1)):
        raise VdnError("libvdn_render.so ABI version does not match include/vdn_render.h; rebuild")
    _lib = lib
    return lib


def call(fn_name, *args):
    """Invoke an entry point; struct args are passed by reference; raises on a non-zero status."""
    lib = load()
    cargs = [ctypes.byref(a) if isinstance(a, ctypes.Structure) else a for a in args]
    rc = getattr(lib, fn_name)(*cargs)
    if rc != 0:
        raise VdnError("%s failed with status %d (%s)" % (fn_name, rc, "argument error" if rc < 0 else "hipError_t"))


def try_call(fn_name, *args, unsupported=-10):
    """call() for an entry point that may decline a shape: False (nothing launched) on the status `unsupported`."""
    lib = load()
    cargs = [ctypes.byref(a) if isinstance(a, ctypes.Structure) else a for a in args]
    rc = getattr(lib, fn_name)(*cargs)
    if rc == unsupported:
        return False
    if rc != 0:
        raise VdnError("%s failed with status %d (%s)" % (fn_name, rc, "argument error" if rc < 0 else "hipError_t"))
    return True


def call_value(fn_name, *args):
    """Invoke an entry point whose int return is a value, not a status (vdn_abi_version, vdn_dw_entry_wgs_*)."""
    return int(getattr(load(), fn_name)(*args))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# ---- host-side helpers on the launch path (each is called tens of times per step) -------------------------------------
import torch  # noqa: E402

try:
    _raw_stream, _get_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice
except AttributeError:          # (an interpreter without the private accessors: the public, slower route)
    _raw_stream = _get_device = None


def stream_handle():
    """The HIP stream handle of torch's current stream on the current device (torch.cuda.current_stream().cuda_stream costs
    ~10 us of Python per call; the step asks 17 times)."""
    if _raw_stream is None:
        return torch.cuda.current_stream().cuda_stream
    return _raw_stream(_get_device())


def module_params(m, out=None):
    """list(m.parameters()) for modules that share no parameters - the same pre-order walk without named_members' generators and
    de-duplication set (8 us -> 1 us per module; render() walks all networks several times per call)."""
    if out is None:
        out = []
    for q in m._parameters.values():
        if q is not None:
            out.append(q)
    for c in m._modules.values():
        if c is not None:
            module_params(c, out)
    return out


def first_param(m):
    """next(m.parameters()) without the generators."""
    for q in m._parameters.values():
        if q is not None:
            return q
    for c in m._modules.values():
        if c is not None:
            q = first_param(c)
            if q is not None:
                return q
    return None


_STREAM_OBJECTS = {}


def current_stream():
    """torch.cuda.current_stream() from a cache keyed by the raw handle (the constructor costs ~10 us; events are recorded on /
    waited for by the current stream half a dozen times per step)."""
    if _raw_stream is None:
        return torch.cuda.current_stream()
    d = _get_device()
    key = (d, _raw_stream(d))                   # (the default stream's handle is 0 on every device)
    s = _STREAM_OBJECTS.get(key)
    if s is None:
        s = _STREAM_OBJECTS[key] = torch.cuda.current_stream()
    return s
