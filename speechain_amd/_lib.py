"""ctypes binding of libspeechain_amd.so -- the C ABI declared in include/speechain_amd.h.

The header is the single source of truth: argument types are parsed from it at load time, so the Python side
cannot drift from the declarations.  There is NO fallback: if the library is missing or a call fails the caller
gets an exception.
"""
import ctypes
import os
import re
from typing import Dict, List, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SC_LIB_PATH") or os.path.join(_HERE, "lib", "libspeechain_amd.so")  # (SC_LIB_PATH: an instrumented build, scripts/*_probe.py)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "speechain_amd.h")

_SCALARS = {"int": ctypes.c_int, "long long": ctypes.c_longlong, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
            "unsigned long long": ctypes.c_ulonglong, "hipStream_t": ctypes.c_void_p}
_DECL = re.compile(r"^\s*(const char\*|int|long long|void)\s+(sc_\w+)\s*\(([^;{]*)\)\s*;", re.M | re.S)


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[str, List[Tuple[str, str]]]]:
    """{symbol: (return type, [(ctype string, arg name), ...])} for every `sc_*` prototype in the header."""
    text = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
    out = {}
    for ret, name, args in _DECL.findall(text):
        parsed = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                m = re.match(r"^(.*?)(\w+)$", a)
                parsed.append((m.group(1).strip(), m.group(2)))
        out[name] = (ret, parsed)
    return out


def _ctype(t: str):
    if "*" in t:
        return ctypes.c_void_p
    t = t.replace("const ", "").strip()
    return _SCALARS[t]


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = None
        # optional per-entry-point HIP-event timing (bench.py): {symbol: [(start, stop), ...]} recorded on the
        # current stream tightly around the launch; None = off (the default, zero overhead)
        self.timing = None
        self.flop_log = None  # optional list receiving 2*M*N*K*batch of every sc_gemm call (bench.py roofline)
        self.kernel_log = None  # optional list receiving sc_gemm_last_kernel() after every sc_gemm / sc_gemm_splitk call
        self.shape_log = None  # optional list receiving (M, N, K, batch) of the same calls

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"speechain_amd: HIP library not built ({LIB_PATH} missing). Run `python -m speechain_amd.build` "
                f"(or __graft_entry__.build()). There is no CPU/PyTorch fallback for the kernels.")
        dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, (ret, args) in self.protos.items():
            fn = getattr(dll, name)  # AttributeError if the header declares something the library lacks
            fn.restype = {"const char*": ctypes.c_char_p, "int": ctypes.c_int, "long long": ctypes.c_longlong,
                          "void": None}[ret]
            fn.argtypes = [_ctype(t) for t, _ in args]
        self._dll = dll
        return dll

    def call(self, name: str, *args):
        """Calls an int-status entry point; raises RuntimeError(sc_last_error()) on a non-zero status."""
        dll = self.load()
        if self.flop_log is not None:
            if name == "sc_gemm":
                self.flop_log.append(2.0 * args[6] * args[7] * args[8] * args[12] * args[13])
            elif name == "sc_gemm_splitk":
                self.flop_log.append(2.0 * args[5] * args[6] * args[7])
        rec = self.timing.get(name) if self.timing is not None else None
        if rec is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = getattr(dll, name)(*args)
            e1.record()
            rec.append((e0, e1))
            if self.kernel_log is not None and name == "sc_wgrad_grouped":
                self.kernel_log.append(13 if dll.sc_wgrad_grouped_last_tile() == 256 else 14)
                if self.shape_log is not None:
                    self.shape_log.append((int(sum(args[5])), int(max(args[6])), int(max(args[7])), args[0]))
            if self.kernel_log is not None and name in ("sc_gemm", "sc_gemm_splitk"):
                self.kernel_log.append(dll.sc_gemm_last_kernel())
                if self.shape_log is not None:
                    self.shape_log.append((args[6], args[7], args[8], args[12] * args[13]) if name == "sc_gemm" else (args[5], args[6], args[7], 1))
        else:
            rc = getattr(dll, name)(*args)
        if rc != 0:
            raise RuntimeError(f"{name} failed ({rc}): {dll.sc_last_error().decode()}")

    def value(self, name: str, *args):
        return getattr(self.load(), name)(*args)


lib = _Lib()


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    """Raw handle of the current HIP stream of the current device (the two C calls below: ~0.3 us; ``torch.cuda.current_stream().cuda_stream``
    builds a Python stream object on every call: ~2.5 us, once per kernel launch on the host-bound paths)."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def require_gpu(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"speechain_amd: {what} runs only on an MI355X device tensor (got device={t.device}); "
                           f"there is no CPU fallback by design.")
