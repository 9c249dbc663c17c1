"""ctypes binding of libfwair_hip.so.  Prototypes are parsed from include/fwair.h (the single source of
truth for the C ABI), so the header, the library and this loader cannot drift apart silently."""
import ctypes
import os
import re

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, 'libfwair_hip.so')
# the header ships inside the package (build.sh copies include/fwair.h next to this file); a source checkout that has not been
# built yet falls back to the repository's include/ directory
HEADER_PATH = next((p for p in (os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fwair.h'),
                                os.path.join(os.path.dirname(_PKG), 'include', 'fwair.h')) if os.path.exists(p)),
                   os.path.join(os.path.dirname(_PKG), 'include', 'fwair.h'))

F32, BF16 = 0, 1
_CT = {'int': ctypes.c_int, 'long': ctypes.c_long, 'float': ctypes.c_float}


def parse_header(path=HEADER_PATH):
    """-> {name: [(ctype, is_pointer), ...]} for every `int fw_*(...)` declaration."""
    txt = open(path).read()
    txt = re.sub(r'/\*.*?\*/', ' ', txt, flags=re.S)
    protos = {}
    for m in re.finditer(r'\bint\s+(fw_\w+)\s*\(([^)]*)\)\s*;', txt):
        name, args = m.group(1), m.group(2).strip()
        sig = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                if '*' in a:
                    sig.append((ctypes.c_void_p, True))
                else:
                    t = a.rsplit(' ', 1)[0].replace('const', '').strip()
                    sig.append((_CT[t], False))
        protos[name] = sig
    return protos


_lib = None
_protos = None


def lib():
    global _lib, _protos
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'fwair: HIP library not built: {LIB_PATH} (run build.sh / __graft_entry__.build()); '
                               'there is no CPU fallback')
        _protos = parse_header()
        L = ctypes.CDLL(LIB_PATH)
        for name, sig in _protos.items():
            fn = getattr(L, name)          # AttributeError if a declared symbol is not exported
            fn.restype = ctypes.c_int
            fn.argtypes = [t for t, _ in sig]
        _lib = L
    return _lib


def protos():
    lib()
    return _protos


def dt(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.bfloat16:
        return BF16
    raise TypeError(f'fwair: unsupported dtype {dtype}')


def _ptr(x):
    if x is None:
        return None
    if torch.is_tensor(x):
        if not x.is_cuda:
            raise RuntimeError('fwair: tensor is not on a HIP device (the product path has no CPU implementation)')
        return x.data_ptr()
    return int(x)


def call(name, *args):
    """Call a C-ABI entry on torch's current HIP stream (appended as the last argument)."""
    L = lib()
    sig = _protos[name]
    if len(args) + 1 != len(sig):
        raise TypeError(f'{name}: expected {len(sig) - 1} arguments, got {len(args)}')
    conv = []
    for a, (t, isptr) in zip(args, sig):
        conv.append(_ptr(a) if isptr else a)
    conv.append(torch.cuda.current_stream().cuda_stream)
    rc = getattr(L, name)(*conv)
    if rc != 0:
        raise RuntimeError(f'fwair: {name} failed with code {rc} '
                           f'({"argument check at source line " + str(-rc) if rc < 0 else "hipError"})')
