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
    "csam_layernorm": [_P, _P, _L, _I, _P, _L, _I, _P, _P, _I, _I, _F],
    "csam_sam_im2col": [_P, _P, _I, _I, _P, _P, _P],
    "csam_dino_im2col": [_P, _P, _I, _I, _P, _P, _P],
    "csam_im2col3x3": [_P, _P, _P, _I],
    "csam_add_cast": [_P, _P, _P, _L, _P, _P, _L, _I],
    "csam_win_attn": [_P, _P, _P, _P, _P, _P, _I, _I, _F],
    "csam_relpos_tables": [_P, _P, _L, _P, _P, _P, _P, _I, _F],
    "csam_flash_attn": [_P, _P, _L, _I, _I, _I, _P, _P, _P, _L, _I, _I, _F],
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


# ----------------------------------------------------------------------------------------------
# encoder-side kernels
# ----------------------------------------------------------------------------------------------
def layernorm(x, gamma, beta, eps, out=None, out_dtype=torch.float16, M=None):
    """Row LayerNorm over the last dim (one wave per row)."""
    if M is None:
        M = x.shape[0]
    D = x.shape[-1]
    if out is None:
        out = torch.empty((M, D), dtype=out_dtype, device=x.device)
    call("csam_layernorm", _stream(), _ptr(x), x.stride(0), _dt(x.dtype), _ptr(out), out.stride(0),
         _dt(out.dtype), _ptr(gamma), _ptr(beta), M, D, float(eps))
    return out


_MEAN = (_F * 3)(123.675, 116.28, 103.53)   # sam.py:38-39 pixel_mean / pixel_std
_STD = (_F * 3)(58.395, 57.12, 57.375)


def sam_im2col(img_chw, out):
    """Sam.preprocess + 16x16 patch im2col: img f32 [3,h,w] (0..255) -> out f16 [4096,768]."""
    assert img_chw.dtype == torch.float32 and img_chw.is_contiguous()
    call("csam_sam_im2col", _stream(), _ptr(img_chw), img_chw.shape[1], img_chw.shape[2], _MEAN, _STD, _ptr(out))
    return out


def dino_im2col(img_chw, out):
    """preprocess + bilinear 1024->1022 + 14x14 patch im2col: -> out f16 [5329,640]."""
    assert img_chw.dtype == torch.float32 and img_chw.is_contiguous()
    call("csam_dino_im2col", _stream(), _ptr(img_chw), img_chw.shape[1], img_chw.shape[2], _MEAN, _STD, _ptr(out))
    return out


def im2col3x3(x, out, C):
    call("csam_im2col3x3", _stream(), _ptr(x), _ptr(out), C)
    return out


def add_cast(a, b=None, b_row_stride=0, out16=None, out32=None):
    """out = a (+ b broadcast by row stride); a f32 [M,N]; writes f16 and/or f32."""
    M, N = a.shape
    call("csam_add_cast", _stream(), _ptr(a), _ptr(b), b_row_stride, _ptr(out16), _ptr(out32), M, N)


def win_attn(qkv, qkv_bias, rel_h, rel_w, out, D, nH, scale):
    call("csam_win_attn", _stream(), _ptr(qkv), _ptr(qkv_bias), _ptr(rel_h), _ptr(rel_w), _ptr(out), D, nH,
         float(scale))
    return out


def relpos_tables(qkv, rel_h, rel_w, th, tw, nH, scale):
    call("csam_relpos_tables", _stream(), _ptr(qkv), qkv.stride(0), _ptr(rel_h), _ptr(rel_w), _ptr(th), _ptr(tw),
         nH, float(scale))


def flash_attn(qkv, out, T, nH, scale, D, th=None, tw=None):
    """qkv f16 [T, 3*D] laid out [3][nH][64] per row -> out f16 [T, D]."""
    call("csam_flash_attn", _stream(), _ptr(qkv), qkv.stride(0), 0, D, 2 * D, _ptr(th), _ptr(tw), _ptr(out),
         out.stride(0), T, nH, float(scale))
    return out
