"""ctypes binding of libcsam_hip.so -- the C-ABI boundary declared in include/csam.h.

PyTorch is used only as plumbing here: device allocations (``torch.empty``), the current HIP
stream handle and ``data_ptr()``.  Every hot operator below is a hand-written HIP kernel for
gfx950.  There is NO CPU / eager fallback: if the library is missing or a call fails, a
``RuntimeError`` is raised (the oracle under ``oracle/`` is test infrastructure and is never
imported from here).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcsam_hip.so")

ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
DT_F16, DT_F32 = 0, 1

_c = ctypes
_P, _I, _L, _F = _c.c_void_p, _c.c_int, _c.c_long, _c.c_float

# name -> argtypes (all return int).  Must match include/csam.h line by line.
SIGNATURES = {
    "csam_gemm_f16": [_P, _P, _L, _P, _L, _P, _L, _I, _P, _P, _P, _L, _I, _I, _I, _I, _I],
}

_lib = None


def lib():
    """Load the library once.  Raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m crowdsam_amd.build` "
                "(the HIP extension is mandatory; there is no CPU fallback)")
        L = ctypes.CDLL(LIB_PATH)
        L.csam_abi_version.restype = _I
        L.csam_last_error.restype = _c.c_char_p
        for name, argt in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argt
            fn.restype = _I
        _lib = L
    return _lib


def _check(rc, name):
    if rc != 0:
        msg = lib().csam_last_error().decode()
        raise RuntimeError(f"{name} failed (rc={rc}): {msg}")


def _ptr(t):
    return None if t is None else _P(t.data_ptr())


def _stream():
    return _P(torch.cuda.current_stream().cuda_stream)


def _dt(dtype):
    if dtype == torch.float16:
        return DT_F16
    if dtype == torch.float32:
        return DT_F32
    raise ValueError(f"unsupported dtype {dtype}")


def call(name, *args):
    rc = getattr(lib(), name)(*args)
    _check(rc, name)


# ----------------------------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------------------------
def gemm_f16(a, w, out=None, bias=None, act=ACT_NONE, residual=None, colscale=None,
             out_dtype=torch.float16, M=None):
    """out[M,N] = act(a[M,K] @ w[N,K]^T + bias) * colscale + residual  (fp16 in, fp32 accumulate)."""
    assert a.dtype == torch.float16 and w.dtype == torch.float16
    assert a.stride(-1) == 1 and w.stride(-1) == 1
    if M is None:
        M = a.shape[0]
    K = a.shape[1]
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.stride(-1) == 1
    call("csam_gemm_f16", _stream(), _ptr(a), a.stride(0), _ptr(w), w.stride(0),
         _ptr(out), out.stride(0), _dt(out.dtype), _ptr(bias), _ptr(colscale),
         _ptr(residual), 0 if residual is None else residual.stride(0),
         DT_F16 if residual is None else _dt(residual.dtype), act, M, N, K)
    return out
